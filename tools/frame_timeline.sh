#!/bin/bash
# Run on the GPU box from the repo root: rocprofv3 kernel trace of N Fast frames (device-resident input,
# no per-kernel events); prints the kernel timeline of one frame with the idle gap before every kernel,
# and per-frame busy / gap totals.   usage: tools/frame_timeline.sh VOXEL N_FRAMES SHOW_FRAME [kind]
export TMPDIR=/tmp
R=$PWD
rm -rf /tmp/p_tl
cat > /tmp/tl_run.py <<PY
import sys, torch
sys.path.insert(0, "$R")
from voxblox_amd import capi, scenes
voxel = float("$1"); nf = int("$2")
kind = {"fast": capi.TSDF_FAST, "merged": capi.TSDF_MERGED, "simple": capi.TSDF_SIMPLE}["${4:-fast}"]
gm = capi.Map(voxel, 16, max_blocks=int(8192 * max(1.0, (0.05 / voxel) ** 3)))
gm.set_stream(torch.cuda.current_stream().cuda_stream)
cfg = capi.tsdf_cfg(default_truncation_distance=4 * voxel)
fr = []
for k in range(nf):
    pose, pts, col = scenes.room_frame(k, 100)
    fr.append((pose, torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda(), pts.shape[0]))
for rep in range(2):   # second pass over the same poses: buffers are allocated, map is warm
    for pose, dp, dc, n in fr:
        gm.integrate_device(kind, cfg, pose[0], pose[1], dp.data_ptr(), dc.data_ptr(), n)
torch.cuda.synchronize()
PY
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/p_tl -- python /tmp/tl_run.py > /tmp/tl.log 2>&1
python - "$2" "$3" <<'PY'
import csv, glob, re, sys
nf = int(sys.argv[1]); show = int(sys.argv[2])
f = glob.glob('/tmp/p_tl/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    m = re.search(r'(k_[a-z_0-9]+)', n)
    if m and 'rocprim' not in n: return m.group(1)
    if 'rocprim' in n:
        for key in ('scan_impl', 'init_lookback', 'onesweep', 'histogram', 'transform'):
            if key in n: return 'rp::' + key
        return 'rp::other'
    return n[:30]
frames = []
for r in rows:
    if 'k_reset_call_state' in r['Kernel_Name']: frames.append([])
    if frames: frames[-1].append(r)
frames = frames[-nf:]            # the second pass
for i, fr in enumerate(frames):
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in fr) / 1e3
    span = (int(fr[-1]['End_Timestamp']) - int(fr[0]['Start_Timestamp'])) / 1e3
    gaps = [(int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3 for a, b in zip(fr[:-1], fr[1:])]
    sync = sum(g for a, g in zip(fr[:-1], gaps) if 'k_publish_state' in a['Kernel_Name'])
    print("frame %2d: %3d kernels span %7.1f us busy %7.1f us gaps %6.1f us (after read-backs %6.1f us)" % (i, len(fr), span, busy, sum(gaps), sync))
fr = frames[show]
t0 = int(fr[0]['Start_Timestamp']); prev = None
for r in fr:
    s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    print("%-26s start=%8.1f dur=%7.1f gap=%6.1f" % (short(r['Kernel_Name']), (s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3))
    prev = e
PY
