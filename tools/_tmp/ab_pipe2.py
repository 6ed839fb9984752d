import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
torch.cuda.init()
from voxblox_amd import capi, scenes, multi_gpu
dev = torch.device("cuda", 0)
cfg = capi.tsdf_cfg(default_truncation_distance=0.2)
fr = []
for k in range(25):
    pose, pts, col = scenes.room_frame(k, 100)
    fr.append((pose, torch.from_numpy(pts).to(dev), torch.from_numpy(col).to(dev), pts.shape[0]))
for stride in (1, 2):
    d = capi.Map(0.05, 16, max_blocks=8192)
    d.profile(True, reset=True)
    for i in range(0, 25, stride):
        pose, dp, dc, n = fr[i]
        d.clear_keep_slots()
        d.integrate_device(capi.TSDF_FAST, cfg, pose[0], pose[1], dp.data_ptr(), dc.data_ptr(), n)
        c = d.counters()
    tab, calls = d.profile_table()
    print("stride", stride, "calls", calls)
    for name, (n, ms) in sorted(tab.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"  {name:40s} {n / calls:8.1f} {1e3 * ms / n:9.2f} {1e3 * ms / calls:9.1f}")
