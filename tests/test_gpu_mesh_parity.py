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
