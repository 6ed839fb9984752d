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
multi_gpu._FULL_CLEAR = bool(os.environ.get("FULL"))
pm = capi.Map(0.05, 16, max_blocks=8192)
dl = [capi.Map(0.05, 16, max_blocks=8192) for _ in range(2)]
sm = multi_gpu.PipelinedShardedTsdfMap(multi_gpu.GpuBackend(pm, dev), [multi_gpu.GpuBackend(d, dev) for d in dl], 0, 1, device=dev)
for i, (pose, dp, dc, n) in enumerate(fr):
    sm.integrate_shard(capi.TSDF_FAST, cfg, pose[0], pose[1], dp, dc, n)
sm.flush(); sm.close()
torch.cuda.synchronize()
