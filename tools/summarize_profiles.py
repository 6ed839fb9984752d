"""Turns the raw rocprofv3 CSVs (tools/collect_profiles.sh) into the committed summaries:
profiles/<tag>_kernel_stats.md and profiles/<tag>_pmc_hbm_traffic.json."""
import collections, csv, json, re, sys

src, tag, frames_stats, frames_pmc = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])


def short(n):
    m = re.search(r'(k_[a-z_0-9]+)', n)
    if m and 'rocprim' not in n:
        return m.group(1)
    if 'rocprim' in n:
        for key in ('onesweep_iteration', 'onesweep_histograms', 'onesweep_scan', 'histogram', 'radix_sort_block_sort',
                    'merge_sort_block_merge', 'scan_impl', 'transform_impl', 'init_lookback'):
            if key in n:
                return 'rocprim::' + key
        return 'rocprim::other'
    return n[:40]


agg = collections.OrderedDict()
for r in csv.DictReader(open(f'{src}/kernel_stats.csv')):
    a = agg.setdefault(short(r['Name']), [0, 0.0])
    a[0] += int(r['Calls']); a[1] += float(r['TotalDurationNs'])
tot = sum(v[1] for v in agg.values())
lines = [f"# rocprofv3 --kernel-trace --stats — {tag}", "",
         "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 20 --warmup 3 "
         "--no-cpu-baseline --mirror-frames 0 --no-variants`",
         f"({frames_stats} frames of BASELINE configs[1]: Fast integrator, 640x480 room stream, 0.05 m). Aggregated by kernel "
         f"(template instances merged); raw CSV: profiles/{tag}_kernel_stats.csv", "",
         "| kernel | calls | calls/frame | total us | avg us | us/frame | % |", "|---|---|---|---|---|---|---|"]
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append(f"| {k} | {v[0]} | {v[0] / frames_stats:.1f} | {v[1] / 1e3:.1f} | {v[1] / 1e3 / v[0]:.2f} | "
                 f"{v[1] / 1e3 / frames_stats:.1f} | {100 * v[1] / tot:.2f} |")
lines.append("")
lines.append(f"GPU-busy per frame: {tot / 1e3 / frames_stats:.1f} us (kernel time only, under the profiler).")
open(f'profiles/{tag}_kernel_stats.md', 'w').write("\n".join(lines) + "\n")

per = {}
for name, col in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    for r in csv.DictReader(open(f'{src}/pmc_{name}_size_counter_collection.csv')):
        if r['Counter_Name'] != col:
            continue
        d = per.setdefault(short(r['Kernel_Name']), {'launches': 0, 'fetch_kb': 0.0, 'write_kb': 0.0, 'seen': set()})
        d[f'{name}_kb'] += float(r['Counter_Value'])
        if name == 'fetch':
            d['launches'] += 1
out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --steps 5 "
                  "--warmup 1 --no-cpu-baseline --mirror-frames 0 --no-variants (two separate passes)",
       "frames": frames_pmc,
       "units": "rocprofv3 reports KB; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE halves wide coalesced (16 B/lane) "
                "streams; these kernels read 4-8 B/lane, left uncorrected; Infinity-Cache hits are counted",
       "per_frame_bytes": {}}
for k, d in sorted(per.items(), key=lambda kv: -(kv[1]['fetch_kb'] + kv[1]['write_kb'])):
    out["per_frame_bytes"][k] = {"launches_per_frame": round(d['launches'] / frames_pmc, 2),
                                 "fetch_bytes": round(d['fetch_kb'] * 1024 / frames_pmc),
                                 "write_bytes": round(d['write_kb'] * 1024 / frames_pmc)}
json.dump(out, open(f'profiles/{tag}_pmc_hbm_traffic.json', 'w'), indent=1)
print("\n".join(lines[:22]))
print(json.dumps({k: v for k, v in list(out["per_frame_bytes"].items())[:6]}, indent=1))
