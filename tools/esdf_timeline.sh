#!/bin/bash
# Run on the GPU box from the repo root: rocprofv3 kernel trace of N frames of the configs[3] stream (Fast integration +
# incremental ESDF update per frame); prints the kernel timeline of the ESDF update of one frame with the idle gap
# before every kernel.   usage: tools/esdf_timeline.sh N_FRAMES SHOW_FRAME
export TMPDIR=/tmp
R=$PWD
rm -rf /tmp/p_etl
cat > /tmp/etl_run.py <<PY
import sys, torch
sys.path.insert(0, "$R")
from voxblox_amd import capi, scenes
voxel = 0.05; nf = int("$1")
gm = capi.Map(voxel, 16, max_blocks=8192)
gm.set_stream(torch.cuda.current_stream().cuda_stream)
cfg = capi.tsdf_cfg(default_truncation_distance=4 * voxel)
ec = capi.esdf_cfg(min_distance_m=2 * voxel)
for k in range(nf):
    pose, pts, col = scenes.room_frame(k, 100)
    dp, dc = torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda()
    gm.integrate_device(capi.TSDF_FAST, cfg, pose[0], pose[1], dp.data_ptr(), dc.data_ptr(), pts.shape[0])
    gm.esdf_update(ec, batch=False, clear_updated_flag=True)
    c = gm.counters()
    print("frame", k, "esdf_blocks", c["esdf_blocks"], "sweeps", c["esdf_sweeps"], "relax_blocks", c["esdf_relaxations"])
torch.cuda.synchronize()
PY
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/p_etl -- python /tmp/etl_run.py > /tmp/etl.log 2>&1
grep "^frame" /tmp/etl.log | tail -8
python - "$2" <<'PY'
import csv, glob, re, sys
show = int(sys.argv[1])
f = glob.glob('/tmp/p_etl/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    m = re.search(r'(k_[a-z_0-9]+)', n)
    return m.group(1) if m else n[:30]
upd = []
for r in rows:
    if 'k_esdf_reset_flags' in r['Kernel_Name']: upd.append([])
    if upd and 'k_reset_call_state' in r['Kernel_Name']: upd.append(None)
    if upd and upd[-1] is not None: upd[-1].append(r)
upd = [u for u in upd if u]
for i, u in enumerate(upd):
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in u) / 1e3
    span = (int(u[-1]['End_Timestamp']) - int(u[0]['Start_Timestamp'])) / 1e3
    tiles = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in u if 'k_esdf_tile' in r['Kernel_Name']]
    print("update %2d: %3d kernels span %7.1f us busy %7.1f us; tile launches: %s" % (i, len(u), span, busy, " ".join("%.0f" % t for t in tiles)))
u = upd[show]
t0 = int(u[0]['Start_Timestamp']); prev = None
for r in u:
    s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    print("%-26s start=%8.1f dur=%7.1f gap=%6.1f" % (short(r['Kernel_Name']), (s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3))
    prev = e
PY
