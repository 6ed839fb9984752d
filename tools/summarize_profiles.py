"""Turns the raw rocprofv3 CSVs (tools/collect_profiles.sh) into the committed summaries:
profiles/<tag>_<workload>_kernel_stats.md (+ .csv) and profiles/<tag>_pmc_hbm_traffic.json.
usage: python tools/summarize_profiles.py gpurun_out/profiles_new r02"""
import collections, csv, json, os, re, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
FRAMES = {"fast": 20, "esdf": 20, "merged_cow": 20, "simple": 6, "sensors4": 2}
WHAT = {"fast": "BASELINE configs[1]: Fast integrator, 640x480 room stream, 0.05 m (the driver-shaped run: frames 5..24 timed after 5 warm-up frames)",
        "esdf": "BASELINE configs[3]: Fast + EsdfIntegrator::updateFromTsdfLayer(true) per frame (reference_order = 1, the reference's own result)",
        "merged_cow": "BASELINE configs[2]: Merged integrator, cow-and-lady-like orbit",
        "simple": "Simple integrator on the room stream",
        "sensors4": "BASELINE configs[4] on one GPU: 4 sensors, 0.02 m, shard + merge (one step = 4 frames)"}


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


for name, frames in FRAMES.items():
    f = f'{src}/{name}_kernel_stats.csv'
    if not os.path.exists(f):
        continue
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        a = agg.setdefault(short(r['Name']), [0, 0.0])
        a[0] += int(r['Calls']); a[1] += float(r['TotalDurationNs'])
    tot = sum(v[1] for v in agg.values())
    lines = [f"# rocprofv3 --kernel-trace --stats — {tag} / {name}", "",
             f"{WHAT[name]}; {frames} timed steps + warm-up under the profiler (`tools/collect_profiles.sh`).",
             f"Aggregated by kernel (template instances merged); raw CSV: profiles/{tag}_{name}_kernel_stats.csv", "",
             "| kernel | calls | total us | avg us | % |", "|---|---|---|---|---|"]
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| {k} | {v[0]} | {v[1] / 1e3:.1f} | {v[1] / 1e3 / v[0]:.2f} | {100 * v[1] / tot:.2f} |")
    lines.append("")
    lines.append(f"GPU-busy total: {tot / 1e6:.2f} ms (kernel time only, warm-up steps included).")
    open(f'profiles/{tag}_{name}_kernel_stats.md', 'w').write("\n".join(lines) + "\n")
    shutil.copy(f, f'profiles/{tag}_{name}_kernel_stats.csv')
    print("\n".join(lines[:16]))

per = {}
PMC_WARMUP, PMC_STEPS = 5, 20  # the driver-shaped command; only the timed frames are kept
frames_pmc = PMC_STEPS
for name, col in (('FETCH_SIZE', 'FETCH_SIZE'), ('WRITE_SIZE', 'WRITE_SIZE')):
    f = f'{src}/pmc_{name}_counter_collection.csv'
    if not os.path.exists(f):
        continue
    rows = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == col]
    rows.sort(key=lambda r: int(r.get('Dispatch_Id', 0)))
    frame = -1  # every frame starts with exactly one k_prep_points launch
    for r in rows:
        k = short(r['Kernel_Name'])
        if k == 'k_prep_points':
            frame += 1
        if frame < PMC_WARMUP or frame >= PMC_WARMUP + PMC_STEPS:
            continue
        d = per.setdefault(k, {'launches': 0, 'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0})
        d[col] += float(r['Counter_Value'])
        if name == 'FETCH_SIZE':
            d['launches'] += 1
if per:
    out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py "
                      "--no-cpu-baseline --no-extras --no-host-path --mirror-frames 0 --profile-frames 0 --steps 20 --warmup 5 (two separate passes)",
           "frames": frames_pmc,
           "frames_desc": "frames 5..24 of the stream = the 20 timed steps of the driver's `bench.py --steps 20 --warmup 5` (warm-up frames dropped by dispatch order)",
           "units": "rocprofv3 reports KB; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE halves wide coalesced (16 B/lane) "
                    "streams; these kernels read 4-8 B/lane, left uncorrected; Infinity-Cache hits are counted",
           "per_frame_bytes": {}}
    for k, d in sorted(per.items(), key=lambda kv: -(kv[1]['FETCH_SIZE'] + kv[1]['WRITE_SIZE'])):
        out["per_frame_bytes"][k] = {"launches_per_frame": round(d['launches'] / frames_pmc, 2),
                                     "fetch_bytes": round(d['FETCH_SIZE'] * 1024 / frames_pmc),
                                     "write_bytes": round(d['WRITE_SIZE'] * 1024 / frames_pmc)}
    tot_b = sum(v["fetch_bytes"] + v["write_bytes"] for v in out["per_frame_bytes"].values())
    out["total_bytes_per_frame"] = tot_b
    json.dump(out, open(f'profiles/{tag}_pmc_hbm_traffic.json', 'w'), indent=1)
    print(json.dumps({k: v for k, v in list(out["per_frame_bytes"].items())[:8]}, indent=1), "total/frame", tot_b)
for f in ("bench", "bench_esdf", "bench_merged_cow", "bench_simple", "bench_sensors4_1gpu", "bench_mesh"):
    p = f'{src}/{f}.json'
    if os.path.exists(p) and os.path.getsize(p) > 10:
        shutil.copy(p, f'profiles/{tag}_{f}.json')
for f in ("timeline_0p05", "timeline_0p02", "timeline_esdf"):
    p = f'{src}/{f}.txt'
    if os.path.exists(p) and os.path.getsize(p) > 10:
        shutil.copy(p, f'profiles/{tag}_{f}.txt')
