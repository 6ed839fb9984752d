import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
torch.cuda.init()
from voxblox_amd import capi, scenes
dev = torch.device("cuda", 0)
frames = [scenes.room_frame(k, 100) for k in range(int(sys.argv[1]))]
d = [(p, torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)) for p, a, c in frames]
for warm in (0, 1, 0, 1):
    gm = capi.Map(0.05, 16, max_blocks=8192)
    cfg = capi.tsdf_cfg(default_truncation_distance=0.2)
    ecfg = capi.esdf_cfg(min_distance_m=0.1, reference_order=1)
    tw = 0.0
    if warm:
        t0 = time.perf_counter(); gm.esdf_update(ecfg, batch=False, clear_updated_flag=True); torch.cuda.synchronize(); tw = (time.perf_counter() - t0) * 1e3
    pose, dp, dc = d[0]
    gm.integrate_device(capi.TSDF_FAST, cfg, pose[0], pose[1], dp.data_ptr(), dc.data_ptr(), dp.shape[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter(); gm.esdf_update(ecfg, batch=False, clear_updated_flag=True); t1 = (time.perf_counter() - t0) * 1e3
    print("warm", warm, "empty update ms", round(tw, 1), "first update ms", round(t1, 1), flush=True)
    gm.close()
