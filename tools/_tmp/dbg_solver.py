import os, sys
sys.path.insert(0, "/root/repo")
os.environ["VBX_DEBUG"] = "1"
from voxblox_amd import capi, scenes
gm = capi.Map(0.05, 16, max_blocks=8192)
cfg = capi.tsdf_cfg(default_truncation_distance=0.2)
for k in range(12):
    pose, pts, col = scenes.room_frame(k, 100)
    if k in (6, 11): print("=== frame", k, file=sys.stderr)
    gm.integrate(capi.TSDF_FAST, cfg, pose[0], pose[1], pts, col)
