"""PCIe-inclusive rate: vbx_tsdf_integrate (host pointers) vs vbx_tsdf_integrate_device."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
torch.cuda.init()
from voxblox_amd import capi, scenes
dev = torch.device("cuda", 0)
cfg = capi.tsdf_cfg(default_truncation_distance=0.2)
frames = [scenes.room_frame(k, 100) for k in range(45)]
for mode in ("device", "host"):
    gm = capi.Map(0.05, 16, max_blocks=8192)
    d = [(p, torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)) for p, a, c in frames]
    t = 0.0
    for i, (pose, pts, col) in enumerate(frames):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if mode == "device":
            gm.integrate_device(capi.TSDF_FAST, cfg, pose[0], pose[1], d[i][1].data_ptr(), d[i][2].data_ptr(), pts.shape[0])
        else:
            gm.integrate(capi.TSDF_FAST, cfg, pose[0], pose[1], pts, col)
        torch.cuda.synchronize()
        if i >= 5: t += time.perf_counter() - t0
    n = len(frames) - 5
    print(mode, "ms/frame", round(t / n * 1e3, 4), "Mpoints/s", round(307200 * n / t / 1e6, 2))
