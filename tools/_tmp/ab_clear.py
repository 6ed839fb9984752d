import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
torch.cuda.init()
from voxblox_amd import capi, scenes
dev = torch.device("cuda", 0)
cfg = capi.tsdf_cfg(default_truncation_distance=0.2)
fr = []
for k in range(25):
    pose, pts, col = scenes.room_frame(k, 100)
    fr.append((pose, torch.from_numpy(pts).to(dev), torch.from_numpy(col).to(dev), pts.shape[0]))
for mode in ("full", "keep", "full", "keep"):
    gm = capi.Map(0.05, 16, max_blocks=8192)
    gm.enable_timing(True)
    tot = {}
    t_clear = 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i, (pose, dp, dc, n) in enumerate(fr):
        tc = time.perf_counter()
        if mode == "full": gm.clear()
        else: gm.clear_keep_slots()
        t_clear += time.perf_counter() - tc
        gm.integrate_device(capi.TSDF_FAST, cfg, pose[0], pose[1], dp.data_ptr(), dc.data_ptr(), n)
        if i >= 5:
            for a, b in gm.timing().items(): tot[a] = tot.get(a, 0.0) + b
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(mode, "ms/frame %.3f" % (dt / len(fr) * 1e3), "clear %.3f" % (t_clear / len(fr) * 1e3), {a: round(b / 20, 3) for a, b in tot.items()}, gm.counters()["replay_rounds"])
    gm.close()
