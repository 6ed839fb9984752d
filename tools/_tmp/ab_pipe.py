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
for mode in ("full", "keep", "full", "keep"):
    multi_gpu._FULL_CLEAR = (mode == "full")
    for pipelined in (True, False):
        pm = capi.Map(0.05, 16, max_blocks=8192)
        dl = [capi.Map(0.05, 16, max_blocks=8192) for _ in range(2)]
        for d in dl: d.enable_timing(True)
        if pipelined:
            sm = multi_gpu.PipelinedShardedTsdfMap(multi_gpu.GpuBackend(pm, dev), [multi_gpu.GpuBackend(d, dev) for d in dl], 0, 1, device=dev)
        else:
            sm = multi_gpu.ShardedTsdfMap(multi_gpu.GpuBackend(pm, dev), multi_gpu.GpuBackend(dl[0], dev), 0, 1)
        tot = {}
        for i, (pose, dp, dc, n) in enumerate(fr):
            if i == 5:
                if pipelined: sm.flush()
                torch.cuda.synchronize(); t0 = time.perf_counter()
            sm.integrate_shard(capi.TSDF_FAST, cfg, pose[0], pose[1], dp, dc, n)
            if i >= 5:
                d = dl[i & 1] if pipelined else dl[0]
                for a, b in d.timing().items(): tot[a] = tot.get(a, 0.0) + b
        if pipelined: sm.flush()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(mode, "pipelined" if pipelined else "sequential", "ms/frame %.3f" % (dt * 1e3), {a: round(b / 20, 3) for a, b in tot.items() if a in ("total_ms", "alloc_ms", "solve_ms", "replay_ms")},
              (sm.stats if pipelined else ""))
        if pipelined: sm.close()
