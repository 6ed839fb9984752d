import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
torch.cuda.init()
from voxblox_amd import capi, scenes, multi_gpu
dev = torch.device("cuda", 0)
cfg = capi.tsdf_cfg(default_truncation_distance=0.2)
fr = []
for k in range(16):
    pose, pts, col = scenes.room_frame(k, 100)
    fr.append((pose, torch.from_numpy(pts).to(dev), torch.from_numpy(col).to(dev), pts.shape[0]))
pm = capi.Map(0.05, 16, max_blocks=8192)
dl = [capi.Map(0.05, 16, max_blocks=8192) for _ in range(2)]
for d in dl: d.enable_timing(True)
sm = multi_gpu.PipelinedShardedTsdfMap(multi_gpu.GpuBackend(pm, dev), [multi_gpu.GpuBackend(d, dev) for d in dl], 0, 1, device=dev)
for i, (pose, dp, dc, n) in enumerate(fr):
    t0 = time.perf_counter()
    sm.integrate_shard(capi.TSDF_FAST, cfg, pose[0], pose[1], dp, dc, n)
    t = dl[i & 1].timing(); c = dl[i & 1].counters()
    print(i, "wall %.3f" % ((time.perf_counter() - t0) * 1e3), {a: round(b, 3) for a, b in t.items() if a in ("total_ms", "prep_ms", "alloc_ms", "solve_ms", "replay_ms")}, "blocks_alloc", c["blocks_allocated"])
sm.close()
