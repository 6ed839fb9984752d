# SQ instruction / wait counters of the Fast integrator's kernels (BASELINE configs[1]); separate --pmc passes,
# --kernel-trace only.  Run on the GPU box from the repo root; prints per-launch averages.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_WAVES GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/q_$N -- python $R/tools/kernel_table.py 0.05 20 10 fast > /tmp/qlog_$N.txt 2>&1 || tail -5 /tmp/qlog_$N.txt
  python - "$N" /tmp/q_$N <<'PY'
import csv, glob, sys, collections, re
n, d = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/*/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(f[0])):
    mm = re.search(r"(k_\w+)", row["Kernel_Name"]); k = mm.group(1) if mm else row["Kernel_Name"][:40]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for k in sorted(agg):
    if k.startswith("k_fast_sweep") or k.startswith("k_rsort") or k.startswith("k_scan") or k.startswith("k_strict") or k.startswith("k_fast_build"):
        print(n, k, {c: round(v / cnt[(k, c)]) for c, v in agg[k].items()}, "launches", max(cnt[(k, c)] for c in agg[k]))
PY
done
