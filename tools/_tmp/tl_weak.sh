export TMPDIR=/tmp
R=$PWD
rm -rf /tmp/p_w
cat > /tmp/w_run.py <<PY
import os, sys, time
sys.path.insert(0, "$R")
import torch
torch.cuda.init()
from voxblox_amd import capi, scenes, multi_gpu
dev = torch.device("cuda", 0)
cfg = capi.tsdf_cfg(default_truncation_distance=0.2)
fr = []
for k in range(30):
    pose, pts, col = scenes.room_frame(k, 100)
    fr.append((pose, torch.from_numpy(pts).to(dev), torch.from_numpy(col).to(dev), pts.shape[0]))
pm = capi.Map(0.05, 16, max_blocks=8192)
dl = [capi.Map(0.05, 16, max_blocks=8192) for _ in range(2)]
sm = multi_gpu.PipelinedShardedTsdfMap(multi_gpu.GpuBackend(pm, dev), [multi_gpu.GpuBackend(d, dev, keep_slots=True) for d in dl], 0, 1, device=dev)
for i, (pose, dp, dc, n) in enumerate(fr):
    if i == 10:
        sm.flush(); torch.cuda.synchronize(); t0 = time.perf_counter()
    sm.integrate_shard(capi.TSDF_FAST, cfg, pose[0], pose[1], dp, dc, n)
sm.flush(); torch.cuda.synchronize()
print("ms/frame", (time.perf_counter() - t0) / 20 * 1e3)
sm.close()
PY
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/p_w -- python /tmp/w_run.py > /tmp/w.log 2>&1
grep "ms/frame" /tmp/w.log
python - <<'PY'
import csv, glob, re
f = glob.glob('/tmp/p_w/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    m = re.search(r'(k_[a-z_0-9]+)', n)
    return m.group(1) if m else n[:34]
idx = [i for i, r in enumerate(rows) if 'k_reset_call_state' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]['Start_Timestamp'])
print("frame span %.1f us, %d kernels; previous frame span %.1f" % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3, b - a, (t0 - int(rows[idx[-4]]['Start_Timestamp'])) / 1e3))
# everything from 40 kernels before the frame start (the clear etc.) to the frame end
prev_end = {}
for r in rows[a - 12:b + 2]:
    s = int(r['Start_Timestamp']); e = int(r['End_Timestamp']); q = r.get('Queue_Id', '?')
    gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
    prev_end[q] = e
    print("%-30s q=%s start=%8.1f dur=%7.1f gap_in_queue=%7.1f" % (short(r['Kernel_Name']), q, (s - t0) / 1e3, (e - s) / 1e3, gap))
PY
