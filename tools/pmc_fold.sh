cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02q
for C in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
  N=$(echo $C | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/p_$N -- python $R/tools/kernel_table.py 0.05 6 4 simple > /tmp/log_$N.txt 2>&1 || tail -5 /tmp/log_$N.txt
  python - "$N" /tmp/p_$N <<'PY'
import csv, glob, sys, collections
n, d = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/*/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(f[0])):
    import re
    mm = re.search(r"(k_\w+|rocprim::\w+::\w+)", row["Kernel_Name"]); k = mm.group(1) if mm else row["Kernel_Name"][:40]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for k in agg:
    if "fold" in k or "emit" in k or "sort" in k:
        print(n, k, {c: round(v / cnt[(k, c)]) for c, v in agg[k].items()})
PY
done
