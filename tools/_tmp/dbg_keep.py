import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ["VBX_DEBUG"] = "1"
from voxblox_amd import capi, scenes
cfg = capi.tsdf_cfg(default_truncation_distance=0.2)
for mode in ("keep", "none"):
    d = capi.Map(0.05, 16, max_blocks=8192)
    for i in range(6):
        pose, pts, col = scenes.room_frame(i, 100)
        if mode == "keep": d.clear_keep_slots()
        print("== mode", mode, "frame", i, file=sys.stderr)
        d.integrate(capi.TSDF_FAST, cfg, pose[0], pose[1], pts, col)
