"""What a lower-phase workgroup of k_esdf_tile spends its time on (library built with -DVBX_ESDF_STATS, passed in VBX_LIB):
iterations to the local fixed point, queue entries evaluated, 100 MHz ticks in the tile load and in the loop."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from voxblox_amd import capi, scenes
capi.LIB_PATH = os.environ["VBX_LIB"]
L = capi.lib()
L.vbx_debug_esdf_stats.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
voxel = 0.05
gm = capi.Map(voxel, 16, max_blocks=8192)
cfg = capi.tsdf_cfg(default_truncation_distance=4 * voxel)
ec = capi.esdf_cfg(min_distance_m=2 * voxel)
out = (C.c_ulonglong * 16)()
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    pose, pts, col = scenes.room_frame(k, 100)
    gm.integrate(capi.TSDF_FAST, cfg, pose[0], pose[1], pts, col)
    L.vbx_debug_esdf_stats(out, 1)
    gm.esdf_update(ec, batch=False, clear_updated_flag=True)
    L.vbx_debug_esdf_stats(out, 1)
    o = [int(x) for x in out]
    n = max(o[0], 1)
    print("frame %2d: lower WGs %4d (changed %4d)  iterations mean %.1f max %d  evals mean %.0f max %d  load us mean %.1f max %.1f  loop us mean %.1f max %.1f (compaction %.1f, relax + mark %.1f)"
          % (k, o[0], o[8], o[1] / n, o[2], o[3] / n, o[9], o[4] / n / 100, o[7] / 100, o[5] / n / 100, o[6] / 100, o[10] / n / 100, o[11] / n / 100))
