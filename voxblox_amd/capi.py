"""ctypes binding of libvbx_hip.so (C-ABI: include/vbx_hip.h).

The shared library is built in-tree by __graft_entry__.build() (hipcc, gfx950).  There is
no CPU fallback: importing this module without the library, or creating a map without a
HIP device, raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (VBX_HIP_LIB: another build of the same library, e.g. one compiled with -DRP_EVQ=8 — tools/r05b_validate.sh)
LIB_PATH = os.environ.get("VBX_HIP_LIB") or os.path.join(_HERE, "csrc", "libvbx_hip.so")

VBX_OK = 0
VBX_ERR_INVALID, VBX_ERR_HIP, VBX_ERR_CAPACITY, VBX_ERR_UNSUPPORTED = -1, -2, -3, -4   # include/vbx_hip.h:30-33
TSDF_SIMPLE, TSDF_MERGED, TSDF_FAST = 1, 2, 3
LAYER_TSDF, LAYER_ESDF = 0, 1
UPDATE_MAP, UPDATE_MESH, UPDATE_ESDF = 1, 2, 4

# every symbol include/vbx_hip.h declares
EXPORTED_SYMBOLS = (
    "vbx_tsdf_cfg_default", "vbx_esdf_cfg_default", "vbx_create", "vbx_destroy",
    "vbx_last_error", "vbx_get_map_cfg", "vbx_set_stream", "vbx_set_pool_limit", "vbx_tsdf_integrate", "vbx_tsdf_integrate_device",
    "vbx_esdf_update", "vbx_esdf_reserve", "vbx_esdf_update_blocks", "vbx_esdf_integrator_clear", "vbx_esdf_add_new_robot_position", "vbx_esdf_robot_updated_blocks", "vbx_num_blocks", "vbx_block_indices", "vbx_blocks_updated", "vbx_blocks_new_ordered", "vbx_block_indices_layer_order", "vbx_set_block_order_tracking",
    "vbx_block_download", "vbx_blocks_download", "vbx_host_alloc", "vbx_host_free", "vbx_block_upload", "vbx_blocks_upload", "vbx_block_remove", "vbx_blocks_remove", "vbx_remove_distant_blocks",
    "vbx_clear", "vbx_clear_keep_slots", "vbx_clear_updated", "vbx_blocks_export_sums", "vbx_blocks_merge_sums", "vbx_blocks_serialize", "vbx_blocks_deserialize", "vbx_get_counters", "vbx_selftest_sort", "vbx_selftest_scan", "vbx_fast_reset_counter_get", "vbx_fast_reset_counter_set", "vbx_enable_timing", "vbx_get_timing",
    "vbx_profile_enable", "vbx_profile_reset", "vbx_profile_get",
    "vbx_selftest_unordered_order", "vbx_selftest_index_set_order", "vbx_mesh_cfg_default", "vbx_mesh_generate", "vbx_mesh_blocks", "vbx_mesh_download", "vbx_mesh_device_ptrs")


class MapCfg(C.Structure):
    _fields_ = [("voxel_size", C.c_float), ("voxels_per_side", C.c_uint32),
                ("max_blocks", C.c_uint32)]


class TsdfCfg(C.Structure):
    """TsdfIntegratorBase::Config (tsdf_integrator.h:56-89)."""
    _fields_ = [("default_truncation_distance", C.c_float), ("max_weight", C.c_float),
                ("voxel_carving_enabled", C.c_int32), ("min_ray_length_m", C.c_float),
                ("max_ray_length_m", C.c_float), ("use_const_weight", C.c_int32),
                ("allow_clear", C.c_int32), ("use_weight_dropoff", C.c_int32),
                ("use_sparsity_compensation_factor", C.c_int32),
                ("sparsity_compensation_factor", C.c_float), ("integrator_threads", C.c_int32),
                ("integration_order_mode", C.c_int32), ("enable_anti_grazing", C.c_int32),
                ("start_voxel_subsampling_factor", C.c_float),
                ("max_consecutive_ray_collisions", C.c_int32),
                ("clear_checks_every_n_frames", C.c_int32), ("max_integration_time_s", C.c_float),
                ("merged_bundle_order", C.c_int32), ("fast_observed_set", C.c_int32)]


class MeshCfg(C.Structure):
    """MeshIntegratorConfig (mesh_integrator.h:47-66)."""
    _fields_ = [("use_color", C.c_int32), ("min_weight", C.c_float)]


class EsdfCfg(C.Structure):
    """EsdfIntegrator::Config (esdf_integrator.h:29-78)."""
    _fields_ = [("full_euclidean_distance", C.c_int32), ("max_distance_m", C.c_float),
                ("min_distance_m", C.c_float), ("default_distance_m", C.c_float),
                ("min_diff_m", C.c_float), ("min_weight", C.c_float), ("num_buckets", C.c_int32),
                ("multi_queue", C.c_int32), ("add_occupied_crust", C.c_int32),
                ("clear_sphere_radius", C.c_float), ("occupied_sphere_radius", C.c_float),
                ("reference_order", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in
                ("points", "rays_cast", "voxel_updates", "voxels_touched", "blocks_allocated",
                 "iterations", "esdf_blocks", "esdf_relaxations", "esdf_sweeps", "replay_rounds",
                 "replay_block_rounds", "time_budget_exceeded", "esdf_respeculated", "points_taken",
                 "esdf_order_inexact")]


class Timing(C.Structure):
    _fields_ = [(k, C.c_float) for k in
                ("total_ms", "prep_ms", "alloc_ms", "solve_ms", "emit_ms", "sort_ms", "fold_ms", "replay_ms")]


class VbxError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libvbx_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # If torch is already imported, make it bring up its bundled HIP runtime before ours is
    # dlopen'ed: the other order leaves torch unable to see the GPU in this process.
    import sys
    _t = sys.modules.get("torch")
    if _t is not None:
        try:
            if _t.cuda.is_available():
                _t.cuda.init()
        except Exception:
            pass
    if not os.path.exists(LIB_PATH):
        raise VbxError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; "
                       "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, f32p, u8p, i32p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
    szp = C.POINTER(C.c_size_t)
    sig = {
        "vbx_tsdf_cfg_default": (None, [C.POINTER(TsdfCfg)]),
        "vbx_esdf_cfg_default": (None, [C.POINTER(EsdfCfg)]),
        "vbx_create": (vp, [C.POINTER(MapCfg), C.c_int]),
        "vbx_destroy": (None, [vp]),
        "vbx_last_error": (C.c_char_p, [vp]),
        "vbx_set_stream": (C.c_int, [vp, vp]),
        "vbx_set_pool_limit": (C.c_int, [vp, C.c_uint32]),
        "vbx_get_map_cfg": (C.c_int, [vp, C.POINTER(MapCfg)]),
        "vbx_tsdf_integrate": (C.c_int, [vp, C.c_int, C.POINTER(TsdfCfg), f32p, f32p, f32p, u8p,
                                         C.c_size_t, C.c_int]),
        "vbx_tsdf_integrate_device": (C.c_int, [vp, C.c_int, C.POINTER(TsdfCfg), f32p, f32p, vp, vp,
                                                C.c_size_t, C.c_int]),
        "vbx_esdf_update": (C.c_int, [vp, C.POINTER(EsdfCfg), C.c_int, C.c_int]),
        "vbx_esdf_reserve": (C.c_int, [vp, C.POINTER(EsdfCfg)]),
        "vbx_esdf_add_new_robot_position": (C.c_int, [vp, C.POINTER(EsdfCfg), f32p]),
        "vbx_esdf_update_blocks": (C.c_int, [vp, C.POINTER(EsdfCfg), i32p, C.c_size_t, C.c_int]),
        "vbx_esdf_robot_updated_blocks": (C.c_int, [vp, C.c_int, i32p, C.c_size_t, szp, C.c_int]),
        "vbx_esdf_integrator_clear": (C.c_int, [vp]),
        "vbx_selftest_unordered_order": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32]),
        "vbx_selftest_index_set_order": (C.c_int64, [i32p, C.c_size_t, i32p]),
        "vbx_mesh_cfg_default": (None, [C.POINTER(MeshCfg)]),
        "vbx_mesh_generate": (C.c_int, [vp, C.POINTER(MeshCfg), C.c_int, C.c_int, szp, szp]),
        "vbx_mesh_blocks": (C.c_int, [vp, i32p, C.POINTER(C.c_uint64), C.c_size_t, szp]),
        "vbx_mesh_download": (C.c_int, [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), u8p, C.c_size_t]),
        "vbx_mesh_device_ptrs": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
        "vbx_selftest_sort": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]),
        "vbx_selftest_scan": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_uint32]),
        "vbx_fast_reset_counter_get": (C.c_int64, []),
        "vbx_fast_reset_counter_set": (None, [C.c_int64]),
        "vbx_num_blocks": (C.c_int, [vp, C.c_int, szp]),
        "vbx_block_indices": (C.c_int, [vp, C.c_int, i32p, C.c_size_t, szp]),
        "vbx_blocks_updated": (C.c_int, [vp, C.c_int, C.c_int, i32p, C.c_size_t, szp]),
        "vbx_blocks_new_ordered": (C.c_int, [vp, i32p, C.c_size_t, szp]),
        "vbx_set_block_order_tracking": (C.c_int, [vp, C.c_int]),
        "vbx_block_indices_layer_order": (C.c_int, [vp, C.c_int, i32p, C.c_size_t, szp, C.POINTER(C.c_int)]),
        "vbx_block_download": (C.c_int, [vp, C.c_int, i32p, vp, u8p, u8p]),
        "vbx_blocks_download": (C.c_int, [vp, C.c_int, i32p, C.c_size_t, vp, u8p, u8p]),
        "vbx_host_alloc": (vp, [C.c_size_t]),
        "vbx_host_free": (None, [vp]),
        "vbx_block_upload": (C.c_int, [vp, C.c_int, i32p, vp, C.c_uint8, C.c_uint8]),
        "vbx_blocks_upload": (C.c_int, [vp, C.c_int, i32p, C.c_size_t, vp, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]),
        "vbx_block_remove": (C.c_int, [vp, C.c_int, i32p]),
        "vbx_blocks_remove": (C.c_int, [vp, C.c_int, i32p, C.c_size_t]),
        "vbx_remove_distant_blocks": (C.c_int, [vp, C.c_int, f32p, C.c_double]),
        "vbx_clear": (C.c_int, [vp, C.c_int]),
        "vbx_clear_keep_slots": (C.c_int, [vp]),
        "vbx_clear_updated": (C.c_int, [vp, C.c_int, C.c_int]),
        "vbx_blocks_serialize": (C.c_int, [vp, C.c_int, i32p, C.c_size_t, C.POINTER(C.c_uint32), u8p]),
        "vbx_blocks_deserialize": (C.c_int, [vp, C.c_int, i32p, C.c_size_t, C.POINTER(C.c_uint32), u8p]),
        "vbx_blocks_export_sums": (C.c_int, [vp, i32p, C.c_size_t, vp]),
        "vbx_blocks_merge_sums": (C.c_int, [vp, i32p, C.c_size_t, vp, C.c_int, C.c_float, C.c_float]),
        "vbx_get_counters": (C.c_int, [vp, C.POINTER(Counters)]),
        "vbx_enable_timing": (C.c_int, [vp, C.c_int]),
        "vbx_get_timing": (C.c_int, [vp, C.POINTER(Timing)]),
        "vbx_profile_enable": (C.c_int, [vp, C.c_int]),
        "vbx_profile_reset": (C.c_int, [vp]),
        "vbx_profile_get": (C.c_int, [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def tsdf_cfg(**kw):
    c = TsdfCfg()
    lib().vbx_tsdf_cfg_default(C.byref(c))
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def mesh_cfg(**kw):
    c = MeshCfg()
    lib().vbx_mesh_cfg_default(C.byref(c))
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def esdf_cfg(**kw):
    c = EsdfCfg()
    lib().vbx_esdf_cfg_default(C.byref(c))
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


TSDF_VOXEL_DTYPE = np.dtype([("distance", "<f4"), ("weight", "<f4"), ("rgba", "u1", (4,))])
ESDF_VOXEL_DTYPE = np.dtype([("distance", "<f4"), ("observed", "u1"), ("hallucinated", "u1"),
                             ("in_queue", "u1"), ("fixed", "u1"), ("parent", "<i4", (3,))])
assert TSDF_VOXEL_DTYPE.itemsize == 12 and ESDF_VOXEL_DTYPE.itemsize == 20


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Map:
    """One (Layer<TsdfVoxel>, Layer<EsdfVoxel>) pair resident in HBM."""

    def __init__(self, voxel_size, voxels_per_side=16, max_blocks=0, device=0):
        self.L = lib()
        self.voxel_size = np.float32(voxel_size)
        self.vps = int(voxels_per_side)
        cfg = MapCfg(float(self.voxel_size), self.vps, int(max_blocks))
        self._pinned = []
        self.h = self.L.vbx_create(C.byref(cfg), int(device))
        if not self.h:
            raise VbxError(self.L.vbx_last_error(None).decode())

    def set_pool_limit(self, max_blocks_limit):
        """The pool doubles on demand; this bounds it (0 = unbounded)."""
        self._chk(self.L.vbx_set_pool_limit(self.h, int(max_blocks_limit)))

    def close(self):
        if getattr(self, "h", None):
            for p in self._pinned:
                self.L.vbx_host_free(p)
            self._pinned = []
            self.L.vbx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != VBX_OK:
            raise VbxError(f"vbx error {rc}: {self.L.vbx_last_error(self.h).decode()}")

    def set_stream(self, stream_ptr):
        self._chk(self.L.vbx_set_stream(self.h, C.c_void_p(stream_ptr)))

    def integrate(self, kind, cfg, pos, quat_wxyz, points_C, rgba, freespace=False):
        """integratePointCloud with host arrays (tsdf_integrator.h:100-103)."""
        pos = np.ascontiguousarray(pos, np.float32)
        q = np.ascontiguousarray(quat_wxyz, np.float32)
        pts = np.ascontiguousarray(points_C, np.float32)
        col = np.ascontiguousarray(rgba, np.uint8)
        if pts.ndim != 2 or pts.shape[1] != 3 or col.shape != (pts.shape[0], 4):
            raise ValueError("points_C must be (N,3) float32 and rgba (N,4) uint8")  # CHECK_EQ, tsdf_integrator.cc:247
        self._chk(self.L.vbx_tsdf_integrate(self.h, int(kind), C.byref(cfg), _fp(pos), _fp(q), _fp(pts),
                                            col.ctypes.data_as(C.POINTER(C.c_uint8)), pts.shape[0],
                                            int(freespace)))

    def integrate_device(self, kind, cfg, pos, quat_wxyz, d_points_ptr, d_rgba_ptr, n, freespace=False):
        pos = np.ascontiguousarray(pos, np.float32)
        q = np.ascontiguousarray(quat_wxyz, np.float32)
        self._chk(self.L.vbx_tsdf_integrate_device(self.h, int(kind), C.byref(cfg), _fp(pos), _fp(q),
                                                   C.c_void_p(d_points_ptr), C.c_void_p(d_rgba_ptr),
                                                   int(n), int(freespace)))

    def esdf_update(self, cfg, batch=False, clear_updated_flag=True):
        self._chk(self.L.vbx_esdf_update(self.h, C.byref(cfg), int(batch), int(clear_updated_flag)))

    def esdf_reserve(self, cfg):
        """Workspace of the ESDF updates before the first one needs it (EsdfIntegrator's constructor, esdf_integrator.cc:7-21)."""
        self._chk(self.L.vbx_esdf_reserve(self.h, C.byref(cfg)))

    def esdf_update_blocks(self, cfg, indices, incremental=False):
        """EsdfIntegrator::updateFromTsdfBlocks (esdf_integrator.cc:124-302)."""
        idx = np.ascontiguousarray(indices, np.int32).reshape(-1, 3)
        self._chk(self.L.vbx_esdf_update_blocks(self.h, C.byref(cfg), idx.ctypes.data_as(C.POINTER(C.c_int32)),
                                                idx.shape[0], int(incremental)))

    def mesh_generate(self, cfg=None, only_mesh_updated_blocks=True, clear_updated_flag=True, download=True):
        """MeshIntegrator<TsdfVoxel>::generateMesh (mesh_integrator.h:142-195).  Returns
        (block indices [n,3], vertex offsets [n+1], vertices [V,3], normals [V,3], colors [V,4] or None);
        with download=False only (indices, offsets)."""
        cfg = cfg or mesh_cfg()
        nb, nv = C.c_size_t(0), C.c_size_t(0)
        self._chk(self.L.vbx_mesh_generate(self.h, C.byref(cfg), int(only_mesh_updated_blocks),
                                           int(clear_updated_flag), C.byref(nb), C.byref(nv)))
        idx = np.zeros((max(nb.value, 1), 3), np.int32)
        off = np.zeros(nb.value + 1, np.uint64)
        n = C.c_size_t(0)
        self._chk(self.L.vbx_mesh_blocks(self.h, idx.ctypes.data_as(C.POINTER(C.c_int32)),
                                         off.ctypes.data_as(C.POINTER(C.c_uint64)), nb.value, C.byref(n)))
        idx = idx[:nb.value]
        if not download:
            return idx, off
        v = np.zeros((max(nv.value, 1), 3), np.float32)
        nn = np.zeros((max(nv.value, 1), 3), np.float32)
        col = np.zeros((max(nv.value, 1), 4), np.uint8) if cfg.use_color else None
        self._chk(self.L.vbx_mesh_download(self.h, v.ctypes.data_as(C.POINTER(C.c_float)),
                                           nn.ctypes.data_as(C.POINTER(C.c_float)),
                                           col.ctypes.data_as(C.POINTER(C.c_uint8)) if col is not None else None,
                                           nv.value))
        return idx, off, v[:nv.value], nn[:nv.value], (col[:nv.value] if col is not None else None)

    def esdf_integrator_clear(self):
        self._chk(self.L.vbx_esdf_integrator_clear(self.h))

    def esdf_add_new_robot_position(self, cfg, position):
        """EsdfIntegrator::addNewRobotPosition (esdf_integrator.cc:25-92)."""
        p = np.ascontiguousarray(position, np.float32)
        self._chk(self.L.vbx_esdf_add_new_robot_position(self.h, C.byref(cfg), p.ctypes.data_as(C.POINTER(C.c_float))))

    def esdf_robot_updated_blocks(self, order=1, clear=False):
        """updated_blocks_ as the reference-order addNewRobotPosition calls left it: (n, 3) block indices, order 0 =
        insertion sequence, 1 = iteration order of the reference's IndexSet."""
        n = C.c_size_t(0)
        self._chk(self.L.vbx_esdf_robot_updated_blocks(self.h, order, None, 0, C.byref(n), 0))
        out = np.zeros((max(n.value, 1), 3), np.int32)
        self._chk(self.L.vbx_esdf_robot_updated_blocks(self.h, order, out.ctypes.data_as(C.POINTER(C.c_int32)), n.value,
                                                       C.byref(n), 1 if clear else 0))
        return out[:n.value]

    def num_blocks(self, layer=LAYER_TSDF):
        n = C.c_size_t(0)
        self._chk(self.L.vbx_num_blocks(self.h, layer, C.byref(n)))
        return n.value

    def block_indices(self, layer=LAYER_TSDF):
        n = C.c_size_t(0)
        self._chk(self.L.vbx_num_blocks(self.h, layer, C.byref(n)))
        out = np.zeros((max(n.value, 1), 3), np.int32)
        self._chk(self.L.vbx_block_indices(self.h, layer, out.ctypes.data_as(C.POINTER(C.c_int32)),
                                           n.value, C.byref(n)))
        return out[:n.value]

    def blocks_updated(self, mask, layer=LAYER_TSDF):
        n = C.c_size_t(0)
        self._chk(self.L.vbx_blocks_updated(self.h, layer, mask, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 3), np.int32)
        self._chk(self.L.vbx_blocks_updated(self.h, layer, mask, out.ctypes.data_as(C.POINTER(C.c_int32)),
                                            n.value, C.byref(n)))
        return out[:n.value]

    def set_block_order_tracking(self, on):
        self._chk(self.L.vbx_set_block_order_tracking(self.h, int(on)))

    def blocks_new_ordered(self):
        """The last integrate call's new TSDF blocks in the reference's Layer::insertBlock sequence."""
        n = C.c_size_t(0)
        self._chk(self.L.vbx_blocks_new_ordered(self.h, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 3), np.int32)
        self._chk(self.L.vbx_blocks_new_ordered(self.h, out.ctypes.data_as(C.POINTER(C.c_int32)), n.value, C.byref(n)))
        return out[:n.value]

    def block_indices_layer_order(self, mask=0):
        """(indices in the iteration order of the reference's Layer, exact flag)."""
        n = C.c_size_t(0)
        ex = C.c_int(0)
        self._chk(self.L.vbx_block_indices_layer_order(self.h, mask, None, 0, C.byref(n), C.byref(ex)))
        out = np.zeros((max(n.value, 1), 3), np.int32)
        self._chk(self.L.vbx_block_indices_layer_order(self.h, mask, out.ctypes.data_as(C.POINTER(C.c_int32)), n.value,
                                                       C.byref(n), C.byref(ex)))
        return out[:n.value], bool(ex.value)

    def block_download(self, idx, layer=LAYER_TSDF):
        """Returns (structured voxel array in the reference's AoS layout, updated bits, has_data)."""
        idx = np.ascontiguousarray(idx, np.int32)
        dt = TSDF_VOXEL_DTYPE if layer == LAYER_TSDF else ESDF_VOXEL_DTYPE
        out = np.zeros(self.vps ** 3, dt)
        u = C.c_uint8(0)
        hd = C.c_uint8(0)
        self._chk(self.L.vbx_block_download(self.h, layer, idx.ctypes.data_as(C.POINTER(C.c_int32)),
                                            out.ctypes.data_as(C.c_void_p), C.byref(u), C.byref(hd)))
        return out, u.value, hd.value

    def pinned_voxels(self, max_blocks, layer=LAYER_TSDF):
        """Page-locked staging array [max_blocks, vps^3] for blocks_download(out=...); lives as
        long as the Map."""
        dt = TSDF_VOXEL_DTYPE if layer == LAYER_TSDF else ESDF_VOXEL_DTYPE
        nbytes = int(max_blocks) * self.vps ** 3 * dt.itemsize
        p = self.L.vbx_host_alloc(nbytes)
        if not p:
            raise MemoryError("vbx_host_alloc failed")
        self._pinned.append(p)
        buf = (C.c_uint8 * nbytes).from_address(p)
        return np.frombuffer(buf, dtype=dt).reshape(int(max_blocks), self.vps ** 3)

    def blocks_download(self, indices, layer=LAYER_TSDF, out=None):
        """Bulk mirror: (voxels[n, vps^3] in the reference's AoS layout, updated bits[n], has_data[n])."""
        idx = np.ascontiguousarray(indices, np.int32).reshape(-1, 3)
        n = idx.shape[0]
        dt = TSDF_VOXEL_DTYPE if layer == LAYER_TSDF else ESDF_VOXEL_DTYPE
        if out is not None:
            assert out.dtype == dt and out.flags.c_contiguous and out.shape[0] >= n
            out = out[:n]
        else:
            out = np.empty((n, self.vps ** 3), dt)
        u = np.zeros(n, np.uint8)
        hd = np.zeros(n, np.uint8)
        self._chk(self.L.vbx_blocks_download(self.h, layer, idx.ctypes.data_as(C.POINTER(C.c_int32)), n,
                                             out.ctypes.data_as(C.c_void_p), u.ctypes.data_as(C.POINTER(C.c_uint8)),
                                             hd.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out, u, hd

    def block_upload(self, idx, voxels, updated_bits=7, has_data=0, layer=LAYER_TSDF):
        idx = np.ascontiguousarray(idx, np.int32)
        dt = TSDF_VOXEL_DTYPE if layer == LAYER_TSDF else ESDF_VOXEL_DTYPE
        v = np.ascontiguousarray(voxels, dt)
        assert v.shape == (self.vps ** 3,)
        self._chk(self.L.vbx_block_upload(self.h, layer, idx.ctypes.data_as(C.POINTER(C.c_int32)),
                                          v.ctypes.data_as(C.c_void_p), updated_bits, has_data))

    def blocks_upload(self, idx_xyz, voxels, updated_bits, has_data=None, layer=LAYER_TSDF):
        """n blocks at once: idx_xyz (n,3), voxels (n, vps^3) of the layer's AoS dtype, updated_bits (n,) uint8."""
        idx = np.ascontiguousarray(idx_xyz, np.int32).reshape(-1, 3)
        n = idx.shape[0]
        dt = TSDF_VOXEL_DTYPE if layer == LAYER_TSDF else ESDF_VOXEL_DTYPE
        v = np.ascontiguousarray(voxels, dt)
        assert v.shape == (n, self.vps ** 3)
        ub = np.ascontiguousarray(updated_bits, np.uint8).reshape(n)
        hd = None if has_data is None else np.ascontiguousarray(has_data, np.uint8).reshape(n)
        u8p = C.POINTER(C.c_uint8)
        self._chk(self.L.vbx_blocks_upload(self.h, layer, idx.ctypes.data_as(C.POINTER(C.c_int32)), n,
                                           v.ctypes.data_as(C.c_void_p), ub.ctypes.data_as(u8p),
                                           None if hd is None else hd.ctypes.data_as(u8p)))

    def block_remove(self, idx, layer=LAYER_TSDF):
        idx = np.ascontiguousarray(idx, np.int32)
        self._chk(self.L.vbx_block_remove(self.h, layer, idx.ctypes.data_as(C.POINTER(C.c_int32))))

    def blocks_remove(self, idx, layer=LAYER_TSDF):
        """Layer::removeBlock for a list of BlockIndex rows: one pass over the pool."""
        idx = np.ascontiguousarray(idx, np.int32).reshape(-1, 3)
        self._chk(self.L.vbx_blocks_remove(self.h, layer, idx.ctypes.data_as(C.POINTER(C.c_int32)), idx.shape[0]))

    def remove_distant_blocks(self, center, max_distance, layer=LAYER_TSDF):
        c = np.ascontiguousarray(center, np.float32)
        self._chk(self.L.vbx_remove_distant_blocks(self.h, layer, _fp(c), float(max_distance)))

    def clear(self, layer=LAYER_TSDF):
        self._chk(self.L.vbx_clear(self.h, layer))

    def clear_keep_slots(self):
        """vbx_clear(TSDF) that keeps the blocks' pool slots as invisible candidates (scratch / delta maps)."""
        self._chk(self.L.vbx_clear_keep_slots(self.h))

    def clear_updated(self, mask, layer=LAYER_TSDF):
        self._chk(self.L.vbx_clear_updated(self.h, layer, mask))

    def blocks_serialize(self, idx_xyz, layer=LAYER_TSDF):
        """Block::serializeToIntegers for n blocks -> (words [n, vps^3*(3|2)] uint32, has_data [n])."""
        idx = np.ascontiguousarray(idx_xyz, np.int32).reshape(-1, 3)
        wpb = self.vps ** 3 * (3 if layer == LAYER_TSDF else 2)
        words = np.zeros((idx.shape[0], wpb), np.uint32)
        hd = np.zeros(max(idx.shape[0], 1), np.uint8)
        self._chk(self.L.vbx_blocks_serialize(self.h, layer, idx.ctypes.data_as(C.POINTER(C.c_int32)), idx.shape[0],
                                              words.ctypes.data_as(C.POINTER(C.c_uint32)),
                                              hd.ctypes.data_as(C.POINTER(C.c_uint8))))
        return words, hd[:idx.shape[0]]

    def blocks_deserialize(self, idx_xyz, words, has_data=None, layer=LAYER_TSDF):
        idx = np.ascontiguousarray(idx_xyz, np.int32).reshape(-1, 3)
        wpb = self.vps ** 3 * (3 if layer == LAYER_TSDF else 2)
        w = np.ascontiguousarray(words, np.uint32).reshape(idx.shape[0], wpb)
        hd = None if has_data is None else np.ascontiguousarray(has_data, np.uint8)
        self._chk(self.L.vbx_blocks_deserialize(self.h, layer, idx.ctypes.data_as(C.POINTER(C.c_int32)), idx.shape[0],
                                                w.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                None if hd is None else hd.ctypes.data_as(C.POINTER(C.c_uint8))))

    def export_sums(self, idx_xyz, d_out_ptr):
        """vbx_blocks_export_sums: three planes (distance, weight, colour word) per listed block."""
        idx = np.ascontiguousarray(idx_xyz, np.int32).reshape(-1, 3)
        self._chk(self.L.vbx_blocks_export_sums(self.h, idx.ctypes.data_as(C.POINTER(C.c_int32)), idx.shape[0],
                                                C.c_void_p(d_out_ptr)))

    def merge_sums(self, idx_xyz, d_sums_ptr, apply_caps=False, truncation=0.0, max_weight=0.0):
        idx = np.ascontiguousarray(idx_xyz, np.int32).reshape(-1, 3)
        self._chk(self.L.vbx_blocks_merge_sums(self.h, idx.ctypes.data_as(C.POINTER(C.c_int32)), idx.shape[0],
                                               C.c_void_p(d_sums_ptr), int(apply_caps), float(truncation),
                                               float(max_weight)))

    def selftest_sort(self, n, begin_bit, end_bit, seed=0, with_vals=True):
        self._chk(self.L.vbx_selftest_sort(self.h, int(n), int(begin_bit), int(end_bit), int(seed), int(with_vals)))

    def selftest_scan(self, n, seed=0, repeats=1):
        self._chk(self.L.vbx_selftest_scan(self.h, int(n), int(seed), int(repeats)))

    def counters(self):
        c = Counters()
        self._chk(self.L.vbx_get_counters(self.h, C.byref(c)))
        return {k: int(getattr(c, k)) for k, _ in Counters._fields_}

    def enable_timing(self, on=True):
        self._chk(self.L.vbx_enable_timing(self.h, int(on)))

    def profile(self, on=True, reset=False):
        """Per-kernel profile of the following integrate / ESDF / mesh calls (vbx_profile_enable)."""
        if reset:
            self._chk(self.L.vbx_profile_reset(self.h))
        self._chk(self.L.vbx_profile_enable(self.h, int(on)))

    def profile_table(self):
        """{kernel: (launches, total_ms)}, number of API calls profiled."""
        need = C.c_size_t(0)
        calls = C.c_uint64(0)
        self._chk(self.L.vbx_profile_get(self.h, None, 0, C.byref(need), C.byref(calls)))
        buf = C.create_string_buffer(int(need.value) + 16)
        self._chk(self.L.vbx_profile_get(self.h, buf, len(buf), None, None))
        tab = {}
        for line in buf.value.decode().splitlines():
            name, n, ms = line.split("\t")
            tab[name] = (int(n), float(ms))
        return tab, int(calls.value)

    def timing(self):
        t = Timing()
        self._chk(self.L.vbx_get_timing(self.h, C.byref(t)))
        return {k: float(getattr(t, k)) for k, _ in Timing._fields_}

    def tsdf_dict(self):
        """{(bx,by,bz): (dist, weight, rgba, updated_bits)} for every allocated TSDF block."""
        idx = self.block_indices(LAYER_TSDF)
        v, u, _ = self.blocks_download(idx, LAYER_TSDF)
        return {tuple(int(x) for x in i): (v[k]["distance"].copy(), v[k]["weight"].copy(), v[k]["rgba"].copy(), int(u[k]))
                for k, i in enumerate(idx)}
