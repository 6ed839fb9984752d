#!/bin/bash
# Run on the GPU box from the repo root: kernel trace of a few Fast frames; prints, for the last
# frames, the idle gap the GPU sees after every k_publish_state (= one host read-back) and the total
# busy / idle split.
export TMPDIR=/tmp
R=$PWD
rm -rf /tmp/p_gap
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/p_gap -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --mirror-frames 0 --no-variants "$@" > /tmp/gap.log 2>&1
python - <<'PY'
import csv, glob, re
f = glob.glob('/tmp/p_gap/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-1400:]
gaps_sync, gaps_other = [], []
busy = 0
for a, b in zip(rows[:-1], rows[1:]):
    busy += int(a['End_Timestamp']) - int(a['Start_Timestamp'])
    gap = int(b['Start_Timestamp']) - int(a['End_Timestamp'])
    (gaps_sync if 'k_publish_state' in a['Kernel_Name'] else gaps_other).append(gap / 1e3)
span = (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3
import statistics as st
print("kernels %d span %.1f us busy %.1f us (%.0f %%)" % (len(rows), span, busy / 1e3, 100 * busy / 1e3 / span))
print("gap after k_publish_state: n=%d median %.1f us mean %.1f us total %.1f us" % (len(gaps_sync), st.median(gaps_sync), st.mean(gaps_sync), sum(gaps_sync)))
pos = [g for g in gaps_other if g > 0]
print("gap after other kernels:   n=%d median %.2f us mean %.2f us total %.1f us" % (len(gaps_other), st.median(gaps_other), st.mean(gaps_other), sum(pos)))
PY
