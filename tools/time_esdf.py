import sys, time
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import numpy as np, torch
torch.cuda.init()
from voxblox_amd import capi, scenes
import os
if os.environ.get('VBX_LIB'): capi.LIB_PATH = os.environ['VBX_LIB']
dev = torch.device("cuda", 0)
gm = capi.Map(0.05, 16, max_blocks=8192)
cfg = capi.tsdf_cfg(default_truncation_distance=0.2)
ecfg = capi.esdf_cfg(min_distance_m=0.1, full_euclidean_distance=int('--full' in sys.argv))
frames = [scenes.room_frame(k, 100) for k in range(30)]
d = [(p, torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)) for p, a, c in frames]
for timing in (False, True):
    gm.enable_timing(timing)
    ti = te = 0.0
    parts = []
    for i, (pose, dp, dc) in enumerate(d):
        t0 = time.perf_counter()
        gm.integrate_device(capi.TSDF_FAST, cfg, pose[0], pose[1], dp.data_ptr(), dc.data_ptr(), dp.shape[0])
        t1 = time.perf_counter()
        gm.esdf_update(ecfg, batch=False, clear_updated_flag=True)
        t2 = time.perf_counter()
        if i >= 5:
            ti += t1 - t0; te += t2 - t1
            if timing:
                tm = gm.timing(); ev = ev + tm['total_ms'] if 'ev' in dir() else tm['total_ms']; parts.append((round(tm['total_ms'],2), round(tm['prep_ms'],2), round(tm['solve_ms'],2), round(tm['fold_ms'],2), gm.counters()['esdf_sweeps'], round((t2-t1)*1e3,2)))
    n = len(d) - 5
    print(parts[-3:])
    print("timing", timing, "integrate ms", ti / n * 1e3, "esdf ms", te / n * 1e3, gm.timing() if timing else "", gm.counters())
