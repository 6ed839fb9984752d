#!/bin/bash
# Round 5, on the GPU box from the repo root (through gpurun).  rocprofv3 kernel stats of the driver-shaped command, of the
# configs[3] leg in reference order and of configs[4] on whole-sensor bundles; FETCH_SIZE / WRITE_SIZE passes (separate runs,
# --kernel-trace only) of the driver-shaped command and of the reference-order ESDF stream.  Raw CSVs -> gpurun_out/profiles_new/
# (+ pmc_esdf_*), summaries -> profiles/r05_* by tools/summarize_profiles.py and the python block at the end.
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/profiles_new
rm -rf $OUT; mkdir -p $OUT
cd /tmp
COMMON="--no-cpu-baseline --no-extras --no-host-path --mirror-frames 0 --profile-frames 0"
stats() {  # name, bench args...
  local name=$1; shift
  rm -rf /tmp/p_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -- python $R/bench.py $COMMON "$@" --detail-out $OUT/${name}_detail.json > $OUT/${name}_bench.log 2>&1
  cp /tmp/p_$name/*/*kernel_stats.csv $OUT/${name}_kernel_stats.csv
}
stats fast --steps 20 --warmup 5
stats esdf --esdf --steps 20 --warmup 3
stats sensors4 --workload sensors4 --steps 2 --warmup 1
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$C /tmp/pe_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/p_$C -- python $R/bench.py $COMMON --steps 20 --warmup 5 --detail-out /tmp/d.json > $OUT/pmc_$C.log 2>&1
  cp /tmp/p_$C/*/*counter_collection.csv $OUT/pmc_${C}_counter_collection.csv
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pe_$C -- python $R/bench.py $COMMON --esdf --steps 10 --warmup 3 --detail-out /tmp/d.json > $OUT/pmc_esdf_$C.log 2>&1
  cp /tmp/pe_$C/*/*counter_collection.csv $OUT/pmc_esdf_${C}_counter_collection.csv
done
cd $R
python tools/summarize_profiles.py $OUT r05 > $OUT/summary.txt 2>&1
python - <<'PY'
import csv, json, re
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
json.dump(out, open('profiles/r05_pmc_esdf_ref_order.json', 'w'), indent=1)
print(json.dumps(out["per_frame_bytes"], indent=1)[:1500], "total/update", out["total_bytes_per_frame"])
PY
mkdir -p gpurun_out/profiles_r05_summaries; cp profiles/r05_* gpurun_out/profiles_r05_summaries/ 2>/dev/null
rm -f $OUT/pmc_*_counter_collection.csv   # (tens of MB; the summaries are what is kept)
ls -la $OUT gpurun_out/profiles_r05_summaries
