import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
torch.cuda.init()
from voxblox_amd import capi, scenes
dev = torch.device("cuda", 0)
frames = [scenes.room_frame(k, 100) for k in range(6)]
d = [(p, torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)) for p, a, c in frames]
gm = capi.Map(0.05, 16, max_blocks=8192)
cfg = capi.tsdf_cfg(default_truncation_distance=0.2)
ecfg = capi.esdf_cfg(min_distance_m=0.1, reference_order=1)
gm.esdf_reserve(ecfg)
for i, (pose, dp, dc) in enumerate(d):
    gm.integrate_device(capi.TSDF_FAST, cfg, pose[0], pose[1], dp.data_ptr(), dc.data_ptr(), dp.shape[0])
    gm.esdf_update(ecfg, batch=False, clear_updated_flag=True)
tb = []
for r in range(3):
    t0 = time.perf_counter(); gm.esdf_update(ecfg, batch=True, clear_updated_flag=True); tb.append(round((time.perf_counter() - t0) * 1e3, 2))
print("batch", tb, flush=True)
