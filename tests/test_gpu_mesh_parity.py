"""-m gpu: the device mesher (vbx_mesh_generate) against the oracle's
restatement of MeshIntegrator<TsdfVoxel> on identical TSDF layers.  Bar: per block the same
number of vertices in the same order, vertices / normals / colours BIT-exact (the float
expressions of marching_cubes.h and mesh_integrator.h are replayed one to one)."""
import numpy as np
import pytest

from voxblox_amd import scenes

pytestmark = pytest.mark.gpu


def _layers(oracle, kind, voxel, frames, vps=16, **kw):
    from voxblox_amd import capi
    oracle.lib().orc_fast_reset_counter_set(0)
    ocfg = oracle.tsdf_cfg(default_truncation_distance=4 * voxel, integrator_threads=1, **kw)
    gcfg = capi.tsdf_cfg(default_truncation_distance=4 * voxel, **kw)
    om = oracle.OracleMap(voxel, vps)
    oi = om.tsdf_integrator(kind, ocfg)
    gm = capi.Map(voxel, vps, max_blocks=8192)
    k = {"simple": capi.TSDF_SIMPLE, "merged": capi.TSDF_MERGED, "fast": capi.TSDF_FAST}[kind]
    return om, oi, gm, k, gcfg


def _apply(store, result):
    idx, off, v, n, c = result
    for i, b in enumerate(idx):
        a, e = int(off[i]), int(off[i + 1])
        store[tuple(int(x) for x in b)] = dict(vertices=v[a:e], normals=n[a:e],
                                               colors=None if c is None else c[a:e])


def _compare(gpu_store, oracle_meshes, use_color=True):
    assert set(gpu_store.keys()) == set(oracle_meshes.keys())
    total = 0
    for k, o in oracle_meshes.items():
        g = gpu_store[k]
        assert g["vertices"].shape == o["vertices"].shape, (k, g["vertices"].shape, o["vertices"].shape)
        assert np.array_equal(g["vertices"].view(np.uint32), o["vertices"].view(np.uint32)), k
        assert np.array_equal(g["normals"].view(np.uint32), o["normals"].view(np.uint32)), k
        assert np.array_equal(o["indices"], np.arange(o["vertices"].shape[0], dtype=np.uint64)), k
        if use_color:
            assert np.array_equal(g["colors"], o["colors"]), k
        else:
            assert o["colors"].shape[0] == 0 and g["colors"] is None
        total += o["vertices"].shape[0]
    return total


@pytest.mark.parametrize("kind,voxel", [("merged", 0.1), ("fast", 0.05), ("simple", 0.1)])
def test_incremental_mesh_stream_bit_exact(oracle, kind, voxel):
    """generateMesh(only_mesh_updated_blocks = true, clear_updated_flag = true) after every frame:
    the meshes of the re-meshed blocks replace the stored ones, exactly like MeshLayer."""
    frames = [scenes.room_frame(k, 100, f=80.0, width=160, height=120) for k in (0, 6, 12)]
    om, oi, gm, k, gcfg = _layers(oracle, kind, voxel, frames)
    ml = om.mesh_layer()
    store = {}
    total = 0
    for pose, pts, col in frames:
        oi.integrate(pose[0], pose[1], pts, col)
        gm.integrate(k, gcfg, pose[0], pose[1], pts, col)
        ml.generate(True, True)
        res = gm.mesh_generate(None, True, True)
        assert res[0].shape[0] > 0
        _apply(store, res)
        total = _compare(store, ml.as_dict())
        assert gm.blocks_updated(2).shape[0] == 0          # Update::kMesh cleared on every block
    assert total > 3000
    # a second call finds nothing flagged
    idx, off = gm.mesh_generate(None, True, True, download=False)
    assert idx.shape[0] == 0 and off.tolist() == [0]


@pytest.mark.parametrize("cfg", [dict(use_color=0), dict(min_weight=0.05), dict()])
def test_full_remesh_config_variants(oracle, cfg):
    """only_mesh_updated_blocks = false, clear_updated_flag = false: every block of the layer,
    flags untouched; use_color = false leaves Mesh::colors empty; a larger min_weight drops cubes."""
    from voxblox_amd import capi
    frames = [scenes.room_frame(k, 100, f=80.0, width=160, height=120) for k in (0, 9)]
    om, oi, gm, k, gcfg = _layers(oracle, "merged", 0.1, frames)
    for pose, pts, col in frames:
        oi.integrate(pose[0], pose[1], pts, col)
        gm.integrate(k, gcfg, pose[0], pose[1], pts, col)
    ml = om.mesh_layer()
    ml.generate(False, False, use_color=bool(cfg.get("use_color", 1)), min_weight=cfg.get("min_weight", 1e-4))
    flagged = gm.blocks_updated(2).shape[0]
    store = {}
    _apply(store, gm.mesh_generate(capi.mesh_cfg(**cfg), False, False))
    assert len(store) == gm.num_blocks()
    assert _compare(store, ml.as_dict(), use_color=bool(cfg.get("use_color", 1))) > 1000
    assert gm.blocks_updated(2).shape[0] == flagged == gm.num_blocks()


def test_mesh_vps8_and_empty_map(oracle):
    from voxblox_amd import capi
    gm0 = capi.Map(0.1, 8, max_blocks=64)
    idx, off = gm0.mesh_generate(None, False, True, download=False)
    assert idx.shape[0] == 0
    frames = [scenes.room_frame(3, 100, f=80.0, width=160, height=120)]
    om, oi, gm, k, gcfg = _layers(oracle, "merged", 0.1, frames, vps=8)
    for pose, pts, col in frames:
        oi.integrate(pose[0], pose[1], pts, col)
        gm.integrate(k, gcfg, pose[0], pose[1], pts, col)
    ml = om.mesh_layer()
    ml.generate(True, True)
    store = {}
    _apply(store, gm.mesh_generate(None, True, True))
    assert _compare(store, ml.as_dict()) > 1000


def test_mesh_full_frames_configs1(oracle):
    """BASELINE configs[1] size: two full 640x480 frames, Fast integrator at 0.05 m, incremental
    meshing after each — every vertex, normal and colour bit-exact."""
    frames = [scenes.room_frame(k, 100) for k in (0, 1)]
    om, oi, gm, k, gcfg = _layers(oracle, "fast", 0.05, frames)
    ml = om.mesh_layer()
    store = {}
    for pose, pts, col in frames:
        oi.integrate(pose[0], pose[1], pts, col)
        gm.integrate(k, gcfg, pose[0], pose[1], pts, col)
        ml.generate(True, True)
        _apply(store, gm.mesh_generate(None, True, True))
    assert _compare(store, ml.as_dict()) > 50000


def test_mesh_api_errors():
    from voxblox_amd import capi
    import ctypes as C
    gm = capi.Map(0.1, 16, max_blocks=256)
    pose, pts, col = scenes.room_frame(0, 100, f=80.0, width=160, height=120)
    gm.integrate(capi.TSDF_MERGED, capi.tsdf_cfg(default_truncation_distance=0.4), pose[0], pose[1], pts, col)
    nb, nv = C.c_size_t(0), C.c_size_t(0)
    cfg = capi.mesh_cfg(use_color=0)
    assert gm.L.vbx_mesh_generate(gm.h, C.byref(cfg), 1, 0, C.byref(nb), C.byref(nv)) == capi.VBX_OK
    assert nb.value > 0 and nv.value > 0 and nv.value % 3 == 0
    n = C.c_size_t(0)
    # the count is always reported; a short table is a capacity error
    assert gm.L.vbx_mesh_blocks(gm.h, None, None, 0, C.byref(n)) == capi.VBX_OK and n.value == nb.value
    idx = np.zeros((1, 3), np.int32)
    off = np.zeros(2, np.uint64)
    assert gm.L.vbx_mesh_blocks(gm.h, idx.ctypes.data_as(C.POINTER(C.c_int32)), off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                1, C.byref(n)) == capi.VBX_ERR_CAPACITY
    v = np.zeros((nv.value, 3), np.float32)
    assert gm.L.vbx_mesh_download(gm.h, v.ctypes.data_as(C.POINTER(C.c_float)), None, None, nv.value - 1) == capi.VBX_ERR_CAPACITY
    rgba = np.zeros((nv.value, 4), np.uint8)   # colours were not produced by this call
    assert gm.L.vbx_mesh_download(gm.h, None, None, rgba.ctypes.data_as(C.POINTER(C.c_uint8)), nv.value) == capi.VBX_ERR_INVALID
    assert gm.L.vbx_mesh_download(gm.h, v.ctypes.data_as(C.POINTER(C.c_float)), None, None, nv.value) == capi.VBX_OK
    assert np.isfinite(v).all() and np.abs(v).max() < 10.0
    # clear_updated_flag = 0 left the kMesh bits: the same call again gives the same mesh
    nb2, nv2 = C.c_size_t(0), C.c_size_t(0)
    assert gm.L.vbx_mesh_generate(gm.h, C.byref(cfg), 1, 0, C.byref(nb2), C.byref(nv2)) == capi.VBX_OK
    assert (nb2.value, nv2.value) == (nb.value, nv.value)


def test_python_class_mirror(oracle):
    """voxblox_amd.integrator.MeshIntegrator / MeshLayer / Mesh: the reference's class surface in
    Python, filling a host MeshLayer from vbx_mesh_generate."""
    from voxblox_amd import capi
    from voxblox_amd.integrator import MeshIntegrator, MeshLayer
    frames = [scenes.room_frame(k, 100, f=80.0, width=160, height=120) for k in (0, 6)]
    om, oi, gm, k, gcfg = _layers(oracle, "merged", 0.1, frames)
    ml = om.mesh_layer()
    layer = MeshLayer(0.1 * 16)
    integrator = MeshIntegrator(MeshIntegrator.Config(), gm, layer)
    for pose, pts, col in frames:
        oi.integrate(pose[0], pose[1], pts, col)
        gm.integrate(k, gcfg, pose[0], pose[1], pts, col)
        ml.generate(True, True)
        integrator.generateMesh(True, True)
    ref = ml.as_dict()
    assert set(layer.getAllAllocatedMeshes()) == set(ref) == set(layer.getAllUpdatedMeshes())
    for key, o in ref.items():
        m = layer.getMeshPtrByIndex(key)
        assert np.array_equal(m.vertices.view(np.uint32), o["vertices"].view(np.uint32))
        assert np.array_equal(m.normals.view(np.uint32), o["normals"].view(np.uint32))
        assert np.array_equal(m.colors, o["colors"]) and np.array_equal(m.indices, o["indices"])
        assert np.allclose(m.origin, np.asarray(key, np.float32) * np.float32(1.6))
    with pytest.raises(ValueError):
        MeshIntegrator(MeshIntegrator.Config(), None, layer)
