"""Drives tools/esdf_order_model.cc.  usage: esdf_order_model.py [FRAMES] [VOXEL] [verbose]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from voxblox_amd import scenes
L = C.CDLL(os.path.join(ROOT, "tools", "libesdf_order_model.so"))
fp = C.POINTER(C.c_float)
L.eom_create.restype = C.c_void_p
L.eom_create.argtypes = [C.c_float]
L.eom_integrate.argtypes = [C.c_void_p, fp, fp, fp, C.POINTER(C.c_uint8), C.c_size_t]
L.eom_update.argtypes = [C.c_void_p, C.c_int]
L.eom_update_parallel.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
L.eom_update_parallel.restype = C.c_long
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
voxel = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
vb = int(sys.argv[3]) if len(sys.argv) > 3 else 0
kmax = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 20
smax = int(sys.argv[5]) if len(sys.argv) > 5 else 256
max_iters = int(sys.argv[6]) if len(sys.argv) > 6 else 64
h = L.eom_create(voxel)
L.eom_set_mode.argtypes = [C.c_void_p, C.c_int]
L.eom_set_mode(h, int(os.environ.get('MODE', '1')))
for i in range(n):
    pose, pts, col = scenes.room_frame(i, 100)
    pos = np.ascontiguousarray(pose[0], np.float32); q = np.ascontiguousarray(pose[1], np.float32)
    sub = int(os.environ.get('SUB', '1'))
    pts = np.ascontiguousarray(pts[::sub], np.float32); col = np.ascontiguousarray(col[::sub], np.uint8)
    L.eom_integrate(h, pos.ctypes.data_as(fp), q.ctypes.data_as(fp), pts.ctypes.data_as(fp), col.ctypes.data_as(C.POINTER(C.c_uint8)), pts.shape[0])
    print("frame", i, end=": ", flush=True)
    L.eom_update(h, vb)
    L.eom_update_parallel(h, kmax, smax, max_iters)
