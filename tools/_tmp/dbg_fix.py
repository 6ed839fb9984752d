import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from voxblox_amd import capi, scenes
voxel = 0.02
for mode in ("keep", "full", "persistent"):
    gm = capi.Map(voxel, 16, max_blocks=131072)
    gm.set_stream(torch.cuda.current_stream().cuda_stream)
    gm.enable_timing(True)
    cfg = capi.tsdf_cfg(default_truncation_distance=4 * voxel)
    for k in range(5):
        pose, pts, col = scenes.room_frame(k, 100)
        dp, dc = torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda()
        if mode == "keep": gm.clear_keep_slots()
        elif mode == "full": gm.clear()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        gm.integrate_device(capi.TSDF_FAST, cfg, pose[0], pose[1], dp.data_ptr(), dc.data_ptr(), pts.shape[0])
        torch.cuda.synchronize(); t1 = time.perf_counter()
        tm = gm.timing()
        print(mode, k, "wall ms %.2f" % ((t1 - t0) * 1e3), {k2: round(v, 3) for k2, v in tm.items()}, gm.counters()["blocks_allocated"])
