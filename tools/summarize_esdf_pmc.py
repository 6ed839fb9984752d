"""FETCH_SIZE / WRITE_SIZE passes of the reference-order ESDF leg (gpurun_out/profiles_new/pmc_esdf_*) ->
profiles/<tag>_pmc_esdf_ref_order.json (tag = argv[1], default r05).  Run from the repo root (tools/collect_profiles_r05.sh, tools/collect_profiles_r05e.sh)."""
import csv, json, re, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"
WARM, STEPS = 3, 10
per = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open('gpurun_out/profiles_new/pmc_esdf_%s_counter_collection.csv' % name)))
    rows.sort(key=lambda r: int(r.get('Dispatch_Id', 0)))
    upd = -1
    for r in rows:
        if r['Counter_Name'] != name:
            continue
        m = re.search(r'(k_[a-z_0-9]+)', r['Kernel_Name'])
        k = m.group(1) if m else r['Kernel_Name'][:30]
        if k == 'k_esdf_reset_flags':      # every update starts with exactly one
            upd += 1
        esdf = k.startswith(('k_esdf', 'k_rp_', 'k_cls_', 'k_sphere'))
        if not esdf or upd < WARM or upd >= WARM + STEPS:
            continue
        d = per.setdefault(k, {'launches': 0, 'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0})
        d[name] += float(r['Counter_Value'])
        if name == 'FETCH_SIZE':
            d['launches'] += 1
out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --esdf --steps 10 --warmup 3 (reference_order = 1; two separate passes)",
       "frames": STEPS, "frames_desc": "the 10 timed reference-order ESDF updates (the 3 warm-up updates dropped by dispatch order)",
       "units": "rocprofv3 reports KB; Infinity-Cache hits are counted; WRITE_SIZE attributes L2 write-backs of earlier kernels' lines to whoever runs (upper bound)",
       "per_frame_bytes": {}}
for k, d in sorted(per.items(), key=lambda kv: -(kv[1]['FETCH_SIZE'] + kv[1]['WRITE_SIZE'])):
    out["per_frame_bytes"][k] = {"launches_per_frame": round(d['launches'] / STEPS, 2), "fetch_bytes": round(d['FETCH_SIZE'] * 1024 / STEPS),
                                 "write_bytes": round(d['WRITE_SIZE'] * 1024 / STEPS)}
out["total_bytes_per_frame"] = sum(v["fetch_bytes"] + v["write_bytes"] for v in out["per_frame_bytes"].values())
json.dump(out, open('profiles/%s_pmc_esdf_ref_order.json' % TAG, 'w'), indent=1)
print(json.dumps(out["per_frame_bytes"], indent=1)[:1500], "total/update", out["total_bytes_per_frame"])
