"""Per-round statistics of the Fast integrator's observed-set replay (VBX_DEBUG=1 output on stderr).
usage: VBX_DEBUG=1 [VBX_REPLAY_NO_BLOCKS=1] python tools/replay_stats.py VOXEL N_FRAMES"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from voxblox_amd import capi, scenes
voxel = float(sys.argv[1]); nf = int(sys.argv[2])
gm = capi.Map(voxel, 16, max_blocks=int(8192 * max(1.0, (0.05 / voxel) ** 3)))
cfg = capi.tsdf_cfg(default_truncation_distance=4 * voxel)
gm.enable_timing(True)
for k in range(nf):
    pose, pts, col = scenes.room_frame(k, 100)
    t0 = time.perf_counter()
    gm.integrate(capi.TSDF_FAST, cfg, pose[0], pose[1], pts, col)
    print(f"[frame {k}] {1e3*(time.perf_counter()-t0):.2f} ms timing={gm.timing()} counters={gm.counters()}", file=sys.stderr, flush=True)
