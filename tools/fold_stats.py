"""Chunk statistics of k_fold_long (library built with -DVBX_FOLD_STATS, passed in VBX_LIB)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from voxblox_amd import capi, scenes
capi.LIB_PATH = os.environ["VBX_LIB"]
gm = capi.Map(0.05, 16, max_blocks=8192)
cfg = capi.tsdf_cfg(default_truncation_distance=0.2)
for k in range(12):
    pose, pts, col = scenes.room_frame(k, 100)
    gm.integrate(capi.TSDF_SIMPLE, cfg, pose[0], pose[1], pts, col)
    c = gm.counters()
    print(k, "updates", c["voxel_updates"], "voxels", c["voxels_touched"], "long runs", c["esdf_sweeps"], "chunks: identity", c["iterations"],
          "non-identity", c["esdf_blocks"], "of which sequential (case 3)", c["esdf_relaxations"])
