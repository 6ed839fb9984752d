"""Drives frontier_model of tools/window_model.cc.  usage: frontier_model.py VOXEL FRAME PBUDGET [guess_mode guess_const grow verbose sensor]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from voxblox_amd import scenes
L = C.CDLL(os.path.join(ROOT, "tools", "libwindow_model.so"))
fp = C.POINTER(C.c_float)
L.frontier_model.argtypes = [fp, fp, fp, C.c_uint32, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
voxel = float(sys.argv[1]); frame = int(sys.argv[2]); pb = int(sys.argv[3])
gm = int(sys.argv[4]) if len(sys.argv) > 4 else 0
gc = int(sys.argv[5]) if len(sys.argv) > 5 else 8
grow = int(sys.argv[6]) if len(sys.argv) > 6 else 4
vb = int(sys.argv[7]) if len(sys.argv) > 7 else 0
sensor = int(sys.argv[8]) if len(sys.argv) > 8 else -1
if sensor >= 0:
    pose, pts, col = scenes.room_sensor_frame(sensor, frame, 25)
else:
    pose, pts, col = scenes.room_frame(frame, 100)
pos = np.ascontiguousarray(pose[0], np.float32); q = np.ascontiguousarray(pose[1], np.float32)
pts = np.ascontiguousarray(pts, np.float32)
sys.stdout.flush()
rc = L.frontier_model(pos.ctypes.data_as(fp), q.ctypes.data_as(fp), pts.ctypes.data_as(fp), pts.shape[0], voxel, pb, gm, gc, grow, vb)
