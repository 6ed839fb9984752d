export TMPDIR=/tmp
R=$PWD
for M in keep full; do
rm -rf /tmp/p_$M
if [ $M = full ]; then export FULL=1; else unset FULL; fi
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/p_$M -- python $R/tools/_tmp/ab_pipe3.py > /tmp/tl_$M.log 2>&1)
python - $M <<'PY'
import csv, glob, re, sys
f = glob.glob('/tmp/p_%s/*/*kernel_trace.csv' % sys.argv[1])[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    m = re.search(r'(k_[a-z_0-9]+)', n)
    return m.group(1) if m else n[:28]
# last frame's window: from the last k_reset_call_state
idx = [i for i, r in enumerate(rows) if 'k_reset_call_state' in r['Kernel_Name']]
a = idx[-2]; b = idx[-1]
t0 = int(rows[a]['Start_Timestamp'])
print("==", sys.argv[1], "frame span %.1f us, %d kernels" % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3, b - a))
prev = None
for r in rows[a:a + 45]:
    s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    print("%-30s q=%s start=%8.1f dur=%7.1f" % (short(r['Kernel_Name']), r.get('Queue_Id', '?'), (s - t0) / 1e3, (e - s) / 1e3))
PY
done
