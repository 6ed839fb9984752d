#!/usr/bin/env python
"""Generates tests/golden/reference_digests.json by running every scenario of scenarios.py through
the REAL voxblox sources (/root/reference/voxblox, compiled in place over the dependency stand-ins
as oracle/_ref/libvbxref.so — `make -C oracle ref`).  Run in the build container, where
/root/reference exists; the JSON is committed so that the pinning holds wherever the reference
sources or the prebuilt library are absent.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, HERE)

import oracle_py as O  # noqa: E402
import scenarios as S  # noqa: E402


def main():
    if not O.ref_available():
        raise SystemExit("oracle/_ref/libvbxref.so is not built (make -C oracle ref; needs /root/reference)")
    L = O.ref_lib()
    out = {"_generator": "tests/golden/make_golden.py over oracle/_ref/libvbxref.so (reference sources @ /root/reference)",
           "scenarios": {}}
    for name, sc in S.SCENARIOS.items():
        m = S.run_on_oracle_api(O, L, sc)
        rec = {"tsdf": S.digest_tsdf(m.tsdf_dict())}
        if sc.get("esdf") is not None:
            rec["esdf"] = S.digest_esdf(m.esdf_dict())
            rec["esdf_no_parents"] = S.digest_esdf(m.esdf_dict(), with_parents=False)
        if sc.get("mesh") is not None:
            rec["mesh"] = S.digest_mesh(m.mesh.as_dict())
        out["scenarios"][name] = rec
        print(name, rec)
    with open(os.path.join(HERE, "reference_digests.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
