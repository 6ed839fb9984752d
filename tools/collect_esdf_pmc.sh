#!/bin/bash
# Run on the GPU box from the repo root: FETCH_SIZE / WRITE_SIZE of the ESDF kernels per update (configs[3] stream,
# updateFromTsdfLayer(true) after every frame), separate --pmc passes with --kernel-trace only; writes
# profiles/<TAG>_pmc_esdf_traffic.json (what bench.py fills esdf.roofline.traffic from).   usage: tools/collect_esdf_pmc.sh TAG
export TMPDIR=/tmp
R=$PWD
TAG=${1:-r03}
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pe_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pe_$C -- python $R/bench.py --esdf --no-cpu-baseline --no-extras --no-host-path --mirror-frames 0 --profile-frames 0 --esdf-fidelity-frames 0 --steps 20 --warmup 3 > /tmp/pe_$C.log 2>&1
done
cd $R
python - "$TAG" <<'PY'
import csv, glob, json, re, sys
tag = sys.argv[1]
WARM, STEPS = 3, 20
per = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob('/tmp/pe_%s/*/*counter_collection.csv' % name)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r.get('Dispatch_Id', 0)))
    upd = -1
    for r in rows:
        if r['Counter_Name'] != name:
            continue
        m = re.search(r'(k_[a-z_0-9]+)', r['Kernel_Name'])
        k = m.group(1) if m else r['Kernel_Name'][:30]
        if k == 'k_esdf_reset_flags':
            upd += 1
        if not k.startswith('k_esdf') or upd < WARM or upd >= WARM + STEPS:
            continue
        d = per.setdefault(k, {'launches': 0, 'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0})
        d[name] += float(r['Counter_Value'])
        if name == 'FETCH_SIZE':
            d['launches'] += 1
out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --esdf --steps 20 --warmup 3 (two separate passes)",
       "frames": STEPS, "frames_desc": "the 20 timed ESDF updates (the 3 warm-up updates dropped by dispatch order)",
       "units": "rocprofv3 reports KB; Infinity-Cache hits are counted; WRITE_SIZE attributes L2 write-backs of earlier kernels' lines to whoever runs (upper bound)",
       "per_frame_bytes": {}}
for k, d in sorted(per.items(), key=lambda kv: -(kv[1]['FETCH_SIZE'] + kv[1]['WRITE_SIZE'])):
    out["per_frame_bytes"][k] = {"launches_per_frame": round(d['launches'] / STEPS, 2), "fetch_bytes": round(d['FETCH_SIZE'] * 1024 / STEPS),
                                 "write_bytes": round(d['WRITE_SIZE'] * 1024 / STEPS)}
out["total_bytes_per_frame"] = sum(v["fetch_bytes"] + v["write_bytes"] for v in out["per_frame_bytes"].values())
json.dump(out, open('profiles/%s_pmc_esdf_traffic.json' % tag, 'w'), indent=1)
print(json.dumps(out["per_frame_bytes"], indent=1), "total/update", out["total_bytes_per_frame"])
PY
