"""Drives tools/replay_model.cc on frames of the bench streams.  usage: replay_model.py VOXEL FRAME [guess_mode guess_const verbose]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from voxblox_amd import scenes
L = C.CDLL(os.path.join(ROOT, "tools", "libreplay_model.so"))
fp = C.POINTER(C.c_float)
L.model_run.argtypes = [fp, fp, fp, C.c_uint32, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
voxel = float(sys.argv[1]); frame = int(sys.argv[2])
gm = int(sys.argv[3]) if len(sys.argv) > 3 else 0
gc = int(sys.argv[4]) if len(sys.argv) > 4 else 8
vb = int(sys.argv[5]) if len(sys.argv) > 5 else 1
cm, ca, gmul, gadd = [int(x) for x in (sys.argv[6:10] if len(sys.argv) > 9 else (2, 16, 4, 16))]
er = int(sys.argv[10]) if len(sys.argv) > 10 else 0
pose, pts, col = scenes.room_frame(frame, 100)
pos = np.ascontiguousarray(pose[0], np.float32); q = np.ascontiguousarray(pose[1], np.float32)
pts = np.ascontiguousarray(pts, np.float32)
rc = L.model_run(pos.ctypes.data_as(fp), q.ctypes.data_as(fp), pts.ctypes.data_as(fp), pts.shape[0], voxel, gm, gc, vb, cm, ca, gmul, gadd, er, None, None)
print("rc", rc)
