import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from voxblox_amd import capi, scenes
if os.environ.get("VBX_LIB"): capi.LIB_PATH = os.environ["VBX_LIB"]
gm = capi.Map(0.05, 16, max_blocks=8192)
gm.set_stream(torch.cuda.current_stream().cuda_stream)
cfg = capi.tsdf_cfg(default_truncation_distance=0.2)
fr = []
for k in range(65):
    pose, pts, col = scenes.room_frame(k, 100)
    fr.append((pose, torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda(), pts.shape[0]))
for rep in range(2):
    for i, (pose, dp, dc, n) in enumerate(fr):
        if i == 5:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        gm.integrate_device(capi.TSDF_FAST, cfg, pose[0], pose[1], dp.data_ptr(), dc.data_ptr(), n)
    torch.cuda.synchronize()
    if rep == 0:
        print(os.environ.get("VBX_LIB", "default"), "ms/frame %.4f" % ((time.perf_counter() - t0) / 60 * 1e3))
    break
gm.enable_timing(True)
tot = {}
for i, (pose, dp, dc, n) in enumerate(fr[:30]):
    gm.integrate_device(capi.TSDF_FAST, cfg, pose[0], pose[1], dp.data_ptr(), dc.data_ptr(), n)
    for a, b in gm.timing().items(): tot[a] = tot.get(a, 0.0) + b
print({a: round(b / 30, 4) for a, b in tot.items() if a in ("alloc_ms", "total_ms")})
