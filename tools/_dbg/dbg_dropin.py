import os, sys, json
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch; torch.cuda.init()
import oracle_py as O
import scenarios as S
H, R = O.ref_hip_lib(), O.ref_lib()
for name in ("esdf_batch_min_diff0", "esdf_incremental"):
    g = S.run_on_oracle_api(O, H, S.SCENARIOS[name]).esdf_dict()
    r = S.run_on_oracle_api(O, R, S.SCENARIOS[name]).esdf_dict()
    print(name, "blocks", len(g), len(r), "keys equal", set(g) == set(r))
    nd = nf = npar = nu = 0
    for k in r:
        if k not in g: continue
        nd += int((g[k][0].view(np.uint32) != r[k][0].view(np.uint32)).sum())
        nf += int((g[k][1] != r[k][1]).sum())
        npar += int((g[k][2] != r[k][2]).any(axis=-1).sum()) if g[k][2].ndim > 1 else int((g[k][2] != r[k][2]).sum())
        nu += int(g[k][3] != r[k][3])
    print("  dist diffs", nd, "flag diffs", nf, "parent diffs", npar, "updated-bit diffs", nu)
    if nf:
        for k in r:
            d = np.nonzero(g[k][1] != r[k][1])[0]
            if len(d):
                print("  e.g. block", k, "lin", d[:5], "g flags", g[k][1][d[:5]], "r flags", r[k][1][d[:5]], "g d", g[k][0][d[:5]], "r d", r[k][0][d[:5]])
                break
