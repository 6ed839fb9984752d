"""Prints the kernel timeline of the last frame(s) from a rocprofv3 --kernel-trace CSV."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    m = re.search(r'(k_[a-z_0-9]+)', n)
    if m and 'rocprim' not in n: return m.group(1)
    if 'rocprim' in n:
        for key in ('merge_sort_block_merge', 'radix_sort_block_sort', 'scan_impl', 'transform_impl', 'init_lookback', 'onesweep', 'histogram'):
            if key in n: return 'rp::' + key
        return 'rp::other'
    return n[:30]
t0 = None
for r in rows[-n:]:
    s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    if t0 is None: t0 = s
    print("%-30s start=%9.1f us dur=%8.1f us" % (short(r['Kernel_Name']), (s - t0) / 1e3, (e - s) / 1e3))
