"""Experiment: the four sensors of configs[4] integrated into four delta maps one after the other vs concurrently
(one host thread + one HIP stream per map).  usage: try_concurrent_sensors.py [voxel] [steps]"""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
torch.cuda.init()
from voxblox_amd import capi, scenes
voxel = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
cfg = capi.tsdf_cfg(default_truncation_distance=4 * voxel)
maps = [capi.Map(voxel, 16, max_blocks=16384) for _ in range(4)]
frames = {}
for k in range(steps + 1):
    for s in range(4):
        pose, pts, col = scenes.room_sensor_frame(s, k)
        frames[(s, k)] = (pose, torch.from_numpy(pts).to(dev), torch.from_numpy(col).to(dev), pts.shape[0])
torch.cuda.synchronize()
def one(s, k):
    pose, dp, dc, n = frames[(s, k)]
    maps[s].clear()
    maps[s].integrate_device(capi.TSDF_FAST, cfg, pose[0], pose[1], dp.data_ptr(), dc.data_ptr(), n)
for s in range(4): one(s, 0)   # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(1, steps + 1):
    for s in range(4): one(s, k)
torch.cuda.synchronize()
t_seq = (time.perf_counter() - t0) / steps
sys.setswitchinterval(5e-5)
t0 = time.perf_counter()
for k in range(1, steps + 1):
    th = [threading.Thread(target=one, args=(s, k)) for s in range(4)]
    for t in th: t.start()
    for t in th: t.join()
torch.cuda.synchronize()
t_par = (time.perf_counter() - t0) / steps
print(f"voxel {voxel}: sequential {t_seq*1e3:.2f} ms/step, concurrent {t_par*1e3:.2f} ms/step, x{t_seq/t_par:.2f}")
