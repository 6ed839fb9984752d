"""Per-step cost of the sharded (multi-GPU) frame path on one rank with the RCCL calls forced."""
import os, sys, time
sys.path.insert(0, '/root/repo')
os.environ.setdefault("VBX_FORCE_COLLECTIVES", "1")
import numpy as np, torch
import torch.distributed as dist
from voxblox_amd import capi, scenes, multi_gpu
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
gm = capi.Map(0.05, 16, max_blocks=8192); pm = capi.Map(0.05, 16, max_blocks=8192)
for m in (gm, pm): m.set_stream(torch.cuda.current_stream().cuda_stream)
sm = multi_gpu.ShardedTsdfMap(multi_gpu.GpuBackend(pm, dev), multi_gpu.GpuBackend(gm, dev), 0, 1, dist)
cfg = capi.tsdf_cfg(default_truncation_distance=0.2)
frames = [scenes.room_frame(k, 100) for k in range(30)]
d = [(p, torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)) for p, a, c in frames]
T = {}
def tick(name, t0):
    torch.cuda.synchronize(); t = time.perf_counter(); T[name] = T.get(name, 0.0) + (t - t0); return t
for i, (pose, dp, dc) in enumerate(d):
    if i == 5: T.clear()
    torch.cuda.synchronize(); t = time.perf_counter()
    sm.d.clear(); t = tick("clear", t)
    sm.d.integrate(capi.TSDF_FAST, cfg, pose[0], pose[1], dp, dc, dp.shape[0]); t = tick("integrate", t)
    keys = multi_gpu._sort_rows_zyx(sm.d.block_indices()); t = tick("block_indices", t)
    allk = sm._gather_keys(keys); t = tick("gather_keys", t)
    groups, L = multi_gpu.build_layout(allk, 1); t = tick("layout", t)
    sums = sm.d.zeros((L, 6, 4096)); t = tick("zeros", t)
    sm.d.export_sums(groups[0], sums[:groups[0].shape[0]]); t = tick("export", t)
    mine = sm._reduce_scatter(sums, L); t = tick("reduce_scatter", t)
    sm.p.merge_sums(groups[0], mine[:groups[0].shape[0]], False, 0.0, 0.0); t = tick("merge", t)
n = len(d) - 5
print({k: round(v / n * 1e3, 3) for k, v in T.items()}, "blocks", groups[0].shape[0])
dist.destroy_process_group()
