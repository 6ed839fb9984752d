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
orig = multi_gpu.ShardedTsdfMap.exchange_and_merge
def skip(self):
    self.last = {"payload_bytes": 0, "sent_blocks": 0}
for mode in ("exchange", "no exchange", "exchange", "no exchange", "plain"):
    multi_gpu.ShardedTsdfMap.exchange_and_merge = orig if mode == "exchange" else skip
    pm = capi.Map(0.05, 16, max_blocks=8192)
    dl = [capi.Map(0.05, 16, max_blocks=8192) for _ in range(2)]
    for d in dl + [pm]: d.enable_timing(True)
    sm = multi_gpu.PipelinedShardedTsdfMap(multi_gpu.GpuBackend(pm, dev), [multi_gpu.GpuBackend(d, dev, keep_slots=True) for d in dl], 0, 1, device=dev)
    tot = {}
    for i, (pose, dp, dc, n) in enumerate(fr):
        if i == 5:
            sm.flush()
            torch.cuda.synchronize(); t0 = time.perf_counter()
        if mode == "plain":
            pm.integrate_device(capi.TSDF_FAST, cfg, pose[0], pose[1], dp.data_ptr(), dc.data_ptr(), n)
            d = pm
        else:
            sm.integrate_shard(capi.TSDF_FAST, cfg, pose[0], pose[1], dp, dc, n)
            d = dl[i & 1]
        if i >= 5:
            for a, b in d.timing().items(): tot[a] = tot.get(a, 0.0) + b
    sm.flush()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(mode, "ms/frame %.3f" % (dt * 1e3), {a: round(b / 20, 3) for a, b in tot.items() if a in ("total_ms", "prep_ms", "alloc_ms", "solve_ms", "replay_ms")}, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in sm.stats.items()})
    sm.close()
