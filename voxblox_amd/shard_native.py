"""ctypes binding of libvbx_shard.so (include/vbx_shard.h): the C++ / RCCL host path of the multi-GPU
ray-bundle sharding.  voxblox_amd.multi_gpu is the same protocol over torch.distributed."""
import ctypes as C
import os

import numpy as np

from . import capi

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libvbx_shard.so")
ID_BYTES = 128
# every symbol include/vbx_shard.h declares
EXPORTED_SYMBOLS = ("vbx_shard_get_unique_id", "vbx_shard_create", "vbx_shard_destroy", "vbx_shard_last_error",
                    "vbx_shard_begin_step", "vbx_shard_integrate", "vbx_shard_end_step", "vbx_shard_get_stats",
                    "vbx_shard_owner_of", "vbx_shard_add_delta", "vbx_shard_integrate_shards", "vbx_shard_set_pipelined",
                    "vbx_shard_wait")
_lib = None


class Stats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("steps", "sent_blocks", "received_blocks", "payload_bytes")]


def lib():
    global _lib
    if _lib is None:
        capi.lib()   # libvbx_hip.so first (libvbx_shard.so links it by rpath)
        L = C.CDLL(LIB_PATH)
        vp, fp, i32p, u8p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
        sig = {"vbx_shard_get_unique_id": (C.c_int, [u8p]),
               "vbx_shard_create": (vp, [vp, vp, C.c_int, C.c_int, u8p, C.c_int]),
               "vbx_shard_destroy": (None, [vp]),
               "vbx_shard_last_error": (C.c_char_p, [vp]),
               "vbx_shard_begin_step": (C.c_int, [vp]),
               "vbx_shard_integrate": (C.c_int, [vp, C.c_int, C.POINTER(capi.TsdfCfg), fp, fp, vp, vp, C.c_size_t, C.c_int]),
               "vbx_shard_end_step": (C.c_int, [vp, C.c_int, C.c_float, C.c_float]),
               "vbx_shard_get_stats": (C.c_int, [vp, C.POINTER(Stats)]),
               "vbx_shard_owner_of": (C.c_int, [i32p, C.c_int]),
               "vbx_shard_add_delta": (C.c_int, [vp, vp]),
               "vbx_shard_integrate_shards": (C.c_int, [vp, C.c_int, C.POINTER(capi.TsdfCfg), C.c_size_t, fp, fp,
                                                        C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int]),
               "vbx_shard_set_pipelined": (C.c_int, [vp, C.c_int]),
               "vbx_shard_wait": (C.c_int, [vp])}
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def unique_id():
    buf = (C.c_uint8 * ID_BYTES)()
    if lib().vbx_shard_get_unique_id(buf) != 0:
        raise capi.VbxError("ncclGetUniqueId failed")
    return bytes(buf)


def owner_of(idx, world):
    i = np.ascontiguousarray(idx, np.int32).reshape(3)
    return lib().vbx_shard_owner_of(i.ctypes.data_as(C.POINTER(C.c_int32)), int(world))


class NativeShard:
    """persistent / delta: voxblox_amd.capi.Map objects of equal geometry (owned by the caller)."""

    def __init__(self, persistent, delta, rank=0, world=1, comm_id=None, device=0):
        self.L = lib()
        self.p, self.d = persistent, delta
        idbuf = None
        if comm_id is not None:
            assert len(comm_id) == ID_BYTES
            idbuf = (C.c_uint8 * ID_BYTES).from_buffer_copy(comm_id)
        self.h = self.L.vbx_shard_create(persistent.h, delta.h, int(rank), int(world), idbuf, int(device))
        if not self.h:
            raise capi.VbxError(self.L.vbx_shard_last_error(None).decode())

    def _chk(self, rc):
        if rc != 0:
            raise capi.VbxError(f"vbx_shard error {rc}: {self.L.vbx_shard_last_error(self.h).decode()}")

    def begin_step(self):
        self._chk(self.L.vbx_shard_begin_step(self.h))

    def integrate(self, kind, cfg, pos, quat, d_points_ptr, d_rgba_ptr, n, freespace=False):
        pos = np.ascontiguousarray(pos, np.float32)
        quat = np.ascontiguousarray(quat, np.float32)
        fp = C.POINTER(C.c_float)
        self._chk(self.L.vbx_shard_integrate(self.h, int(kind), C.byref(cfg), pos.ctypes.data_as(fp), quat.ctypes.data_as(fp),
                                             C.c_void_p(int(d_points_ptr)), C.c_void_p(int(d_rgba_ptr)), int(n), int(freespace)))

    def add_delta(self, delta):
        """A further delta map (capi.Map): shard i of a step goes into delta i % n, the deltas run concurrently."""
        self._more = getattr(self, "_more", []) + [delta]   # keep it alive
        self._chk(self.L.vbx_shard_add_delta(self.h, delta.h))

    def set_pipelined(self, on=True):
        self._chk(self.L.vbx_shard_set_pipelined(self.h, int(on)))

    def wait(self):
        self._chk(self.L.vbx_shard_wait(self.h))

    def integrate_shards(self, kind, cfg, shards, freespace=False):
        """shards: [(pos, quat, d_points_ptr, d_rgba_ptr, n)] — all shards of this rank's step, integrated concurrently."""
        n = len(shards)
        pos = np.ascontiguousarray(np.stack([np.asarray(s[0], np.float32) for s in shards]) if n else np.zeros((0, 3)), np.float32)
        quat = np.ascontiguousarray(np.stack([np.asarray(s[1], np.float32) for s in shards]) if n else np.zeros((0, 4)), np.float32)
        pts = (C.c_void_p * max(n, 1))(*[int(s[2]) for s in shards])
        col = (C.c_void_p * max(n, 1))(*[int(s[3]) for s in shards])
        cnt = (C.c_size_t * max(n, 1))(*[int(s[4]) for s in shards])
        fp = C.POINTER(C.c_float)
        self._chk(self.L.vbx_shard_integrate_shards(self.h, int(kind), C.byref(cfg), n, pos.ctypes.data_as(fp), quat.ctypes.data_as(fp),
                                                    pts, col, cnt, int(freespace)))

    def end_step(self, apply_caps=False, truncation=0.0, max_weight=0.0):
        self._chk(self.L.vbx_shard_end_step(self.h, int(apply_caps), float(truncation), float(max_weight)))

    def stats(self):
        s = Stats()
        self._chk(self.L.vbx_shard_get_stats(self.h, C.byref(s)))
        return {k: int(getattr(s, k)) for k, _ in Stats._fields_}

    def close(self):
        if getattr(self, "h", None):
            self.L.vbx_shard_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
