"""Times the reference-order ESDF update (cfg.reference_order = 1) on the configs[3] stream.
usage: time_esdf_strict.py [FRAMES] ; env VBX_ESDF_REPLAY=0 selects the one-wave form, VBX_RP_STATS=1 prints the replay's counters"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
torch.cuda.init()
from voxblox_amd import capi, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
gm = capi.Map(0.05, 16, max_blocks=8192)
cfg = capi.tsdf_cfg(default_truncation_distance=0.2)
ecfg = capi.esdf_cfg(min_distance_m=0.1, reference_order=1)
frames = [scenes.room_frame(k, 100) for k in range(n)]
d = [(p, torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)) for p, a, c in frames]
gm.enable_timing(True)
if not os.environ.get("VBX_NO_RESERVE"):
    gm.esdf_reserve(ecfg)   # (the integrator's workspace, as bench.py: outside the first update)
ts = []
for i, (pose, dp, dc) in enumerate(d):
    gm.integrate_device(capi.TSDF_FAST, cfg, pose[0], pose[1], dp.data_ptr(), dc.data_ptr(), dp.shape[0])
    t0 = time.perf_counter()
    gm.esdf_update(ecfg, batch=False, clear_updated_flag=True)
    t1 = time.perf_counter()
    ts.append((t1 - t0) * 1e3)
    print("frame", i, "esdf ms wall", round(ts[-1], 2), "events ms", round(gm.timing()['total_ms'], 2), gm.counters()['esdf_sweeps'], flush=True)
print("median ms", float(np.median(ts[2:])))
