#!/bin/bash
# usage: tools/prof_kernels.sh <bench args...>   (run on the GPU box from the repo root)
export TMPDIR=/tmp
R=$PWD
rm -rf /tmp/prof
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --no-cpu-baseline --mirror-frames 0 --no-variants "$@" > /tmp/b.log 2>&1
grep -v "^W2026\|^E2026" /tmp/b.log | tail -1 | cut -c1-200
python - <<'PY'
import csv, glob, re, collections
f = glob.glob('/tmp/prof/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    n = r['Name']
    m = re.search(r'(k_[a-z_0-9]+)', n)
    if m and 'rocprim' not in n: k = m.group(1)
    elif 'rocprim' in n:
        k = 'rp::other'
        for key in ('merge_sort_block_merge', 'radix_sort_block_sort', 'scan_impl', 'transform_impl', 'init_lookback', 'onesweep_iteration', 'onesweep_histograms', 'onesweep_scan', 'histogram'):
            if key in n: k = 'rp::' + key; break
    else: k = n[:32]
    a = agg.setdefault(k, [0, 0.0])
    a[0] += int(r['Calls']); a[1] += float(r['TotalDurationNs'])
tot = sum(v[1] for v in agg.values())
print("%-34s %7s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-34s %7d %12.1f %10.2f %6.2f" % (k, v[0], v[1] / 1e3, v[1] / 1e3 / v[0], 100 * v[1] / tot))
PY
