"""Pins the oracle restatement against the REAL reference sources.

oracle/_ref/libvbxref.so = /root/reference/voxblox's own tsdf_integrator.cc,
integrator_utils.cc, esdf_integrator.cc, neighbor_tools.cc (+ the headers they include)
compiled in place over minimal Eigen/glog/minkindr/protobuf stand-ins (oracle/ref_shims,
recipe oracle/Makefile `ref`).  It exports the same orc_* C API, so the same inputs run
through both and every layer must come out BIT-identical — distances, weights, colours,
flags, parents, updated bits and even the unordered_map block iteration order.

Runs wherever the prebuilt .so is present (it travels to the GPU box) or /root/reference
exists; skipped otherwise.  CPU only.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_py as O
from voxblox_amd import scenes

pytestmark = pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built and no /root/reference")


def _frames(n, w=96, h=72, f=48.0):
    return [scenes.room_frame(5 * k, 100, f=f, width=w, height=h) for k in range(n)]


def _both(kind, frames, voxel=0.1, esdf=None, freespace=False, **cfg):
    out = []
    for L in (O.lib(), O.ref_lib()):
        L.orc_fast_reset_counter_set(0)
        m = O.OracleMap(voxel, 16, L=L)
        c = O.TsdfCfg()
        L.orc_tsdf_cfg_default(C.byref(c))
        c.default_truncation_distance = 4 * voxel
        c.integrator_threads = 1
        for k, v in cfg.items():
            setattr(c, k, v)
        it = m.tsdf_integrator(kind, c)
        e = None
        if esdf is not None:
            ec = O.EsdfCfg()
            L.orc_esdf_cfg_default(C.byref(ec))
            ec.min_distance_m = 2 * voxel
            for k, v in esdf.get("cfg", {}).items():
                setattr(ec, k, v)
            e = m.esdf_integrator(ec)
        for pose, pts, col in frames:
            it.integrate(pose[0], pose[1], pts, col, freespace)
            if e is not None and esdf.get("robot"):
                e.add_new_robot_position(pose[0])  # esdf_server.cc:224
            if e is not None and esdf.get("mode") == "incremental":
                e.update_from_tsdf_layer(True)
        if e is not None and esdf.get("mode") == "batch":
            e.update_from_tsdf_layer_batch()
        out.append(m)
    return out


def _same_tsdf(a, b):
    ia, ib = a.block_indices(0), b.block_indices(0)
    assert np.array_equal(ia, ib), "block iteration order differs (unordered_map order is part of the restatement)"
    assert len(ia) > 0
    for i in ia:
        da, wa, ca, ua = a.tsdf_block(i)
        db, wb, cb, ub = b.tsdf_block(i)
        assert ua == ub
        assert np.array_equal(da.view(np.uint32), db.view(np.uint32)), f"distance bits differ in block {tuple(i)}"
        assert np.array_equal(wa.view(np.uint32), wb.view(np.uint32)), f"weight bits differ in block {tuple(i)}"
        assert np.array_equal(ca, cb), f"colours differ in block {tuple(i)}"


def _same_esdf(a, b):
    ia, ib = a.block_indices(1), b.block_indices(1)
    assert np.array_equal(ia, ib) and len(ia) > 0
    for i in ia:
        da, fa, pa, ua = a.esdf_block(i)
        db, fb, pb, ub = b.esdf_block(i)
        assert ua == ub
        assert np.array_equal(fa, fb), f"flags differ in block {tuple(i)}"
        assert np.array_equal(da.view(np.uint32), db.view(np.uint32)), f"distance bits differ in block {tuple(i)}"
        assert np.array_equal(pa, pb), f"parents differ in block {tuple(i)}"


@pytest.mark.parametrize("kind", ["simple", "merged", "fast"])
def test_tsdf_integrators_bit_identical(kind):
    a, b = _both(kind, _frames(4))
    _same_tsdf(a, b)


@pytest.mark.parametrize("kind,cfg", [
    ("simple", dict(voxel_carving_enabled=0)),
    ("simple", dict(use_const_weight=1, use_weight_dropoff=0)),
    ("simple", dict(use_sparsity_compensation_factor=1, sparsity_compensation_factor=3.0)),
    ("merged", dict(use_sparsity_compensation_factor=1, sparsity_compensation_factor=3.0)),
    ("fast", dict(use_sparsity_compensation_factor=1, sparsity_compensation_factor=3.0)),
    ("simple", dict(allow_clear=0, max_ray_length_m=3.0)),
    ("merged", dict(enable_anti_grazing=1)),
    ("merged", dict(max_ray_length_m=2.5, min_ray_length_m=1.5)),
    ("fast", dict(max_consecutive_ray_collisions=0)),
    ("fast", dict(start_voxel_subsampling_factor=1.0, max_consecutive_ray_collisions=5)),
    ("fast", dict(integration_order_mode=1)),
])
def test_tsdf_config_variants_bit_identical(kind, cfg):
    a, b = _both(kind, _frames(2), **cfg)
    _same_tsdf(a, b)


def test_fast_small_voxels_many_frames():
    a, b = _both("fast", _frames(6), voxel=0.05)
    _same_tsdf(a, b)


def test_fast_approx_set_offset_wrap():
    """ApproxHashSet<20, 10000>::resetApproxSet (approx_hash_array.h:156-169): 9997 empty clouds with
    clear_checks_every_n_frames = 1 walk both sets to offset 9997; the four real frames then run
    at offsets 9998, 9999, 0 (full reset, size_t-max sentinel back in slot 0) and 1."""
    empty = (np.zeros((0, 3), np.float32), np.zeros((0, 4), np.uint8))
    frames = _frames(4)
    pad = [(frames[0][0], empty[0], empty[1])] * 9997
    a, b = _both("fast", pad + frames, voxel=0.05, clear_checks_every_n_frames=1)
    _same_tsdf(a, b)


def test_freespace_points():
    a, b = _both("merged", _frames(2), freespace=True)
    _same_tsdf(a, b)


def test_esdf_incremental_bit_identical():
    a, b = _both("merged", _frames(4), esdf=dict(mode="incremental"))
    _same_tsdf(a, b)
    _same_esdf(a, b)


def test_esdf_batch_and_variants_bit_identical():
    for cfg in (dict(), dict(min_diff_m=0.0, multi_queue=1), dict(add_occupied_crust=1),
                dict(full_euclidean_distance=1)):
        a, b = _both("simple", _frames(2), esdf=dict(mode="batch", cfg=cfg))
        _same_esdf(a, b)


def test_esdf_full_euclidean_incremental():
    a, b = _both("merged", _frames(3), esdf=dict(mode="incremental", cfg=dict(full_euclidean_distance=1)))
    _same_esdf(a, b)


def test_esdf_add_new_robot_position_bit_identical():
    """addNewRobotPosition (clear sphere + occupied sphere) before every incremental update, as
    EsdfServer does with clear_sphere_for_planning (esdf_server.cc:219-226)."""
    for cfg in (dict(clear_sphere_radius=0.6, occupied_sphere_radius=1.6),
                dict(clear_sphere_radius=0.75, occupied_sphere_radius=1.25, min_diff_m=0.0)):
        a, b = _both("merged", _frames(4), esdf=dict(mode="incremental", robot=True, cfg=cfg))
        _same_esdf(a, b)
        n_tsdf, n_esdf = len(a.block_indices(0)), len(a.block_indices(1))
        assert n_esdf > n_tsdf  # the spheres allocate ESDF blocks the TSDF layer does not have
        d, f, p, u = a.esdf_block(a.block_indices(1)[0])
        assert f.any()


def test_esdf_update_from_tsdf_blocks_and_clear_bit_identical():
    """updateFromTsdfBlocks(list, incremental) on block subsets, then addNewRobotPosition +
    clear() (the queued work must be forgotten) + a regular incremental update."""
    out = []
    for L in (O.lib(), O.ref_lib()):
        L.orc_fast_reset_counter_set(0)
        m = O.OracleMap(0.1, 16, L=L)
        c = O.TsdfCfg(); L.orc_tsdf_cfg_default(C.byref(c))
        c.default_truncation_distance = 0.4; c.integrator_threads = 1
        it = m.tsdf_integrator("merged", c)
        for pose, pts, col in _frames(3):
            it.integrate(pose[0], pose[1], pts, col)
        ec = O.EsdfCfg(); L.orc_esdf_cfg_default(C.byref(ec))
        ec.min_distance_m = 0.2; ec.clear_sphere_radius = 0.6; ec.occupied_sphere_radius = 1.4
        e = m.esdf_integrator(ec)
        blocks = m.block_indices(0)
        e.update_from_tsdf_blocks(blocks[::2], False)
        e.update_from_tsdf_blocks(blocks[1::2], True)
        e.add_new_robot_position(np.array([0.3, -0.2, -0.5], np.float32))
        e.clear()
        e.update_from_tsdf_layer(True)
        out.append(m)
    _same_esdf(*out)


def test_helpers_bit_identical():
    rng = np.random.RandomState(3)
    La, Lb = O.lib(), O.ref_lib()
    f32p, i64p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_int64), C.POINTER(C.c_int32)
    for _ in range(300):
        p = (rng.uniform(-20, 20, 3)).astype(np.float32)
        inv = np.float32(rng.choice([20.0, 10.0, 5.0, 1.25, 40.0]))
        oa, ob = np.zeros(3, np.int64), np.zeros(3, np.int64)
        La.orc_grid_index_from_point(p.ctypes.data_as(f32p), float(inv), oa.ctypes.data_as(i64p))
        Lb.orc_grid_index_from_point(p.ctypes.data_as(f32p), float(inv), ob.ctypes.data_as(i64p))
        assert np.array_equal(oa, ob)
        ca, cb = np.zeros(3, np.float32), np.zeros(3, np.float32)
        La.orc_center_point_from_grid_index(oa.ctypes.data_as(i64p), float(1 / inv), ca.ctypes.data_as(f32p))
        Lb.orc_center_point_from_grid_index(oa.ctypes.data_as(i64p), float(1 / inv), cb.ctypes.data_as(f32p))
        assert np.array_equal(ca.view(np.uint32), cb.view(np.uint32))
        assert La.orc_long_index_hash(oa.ctypes.data_as(i64p)) == Lb.orc_long_index_hash(oa.ctypes.data_as(i64p))
        q = rng.normal(size=4); q = (q / np.linalg.norm(q)).astype(np.float32)
        t = rng.uniform(-3, 3, 3).astype(np.float32)
        ta, tb = np.zeros(3, np.float32), np.zeros(3, np.float32)
        La.orc_transform_point(t.ctypes.data_as(f32p), q.ctypes.data_as(f32p), p.ctypes.data_as(f32p), ta.ctypes.data_as(f32p))
        Lb.orc_transform_point(t.ctypes.data_as(f32p), q.ctypes.data_as(f32p), p.ctypes.data_as(f32p), tb.ctypes.data_as(f32p))
        assert np.array_equal(ta.view(np.uint32), tb.view(np.uint32))
        o = rng.uniform(-2, 2, 3).astype(np.float32)
        ba, bb = np.zeros((2048, 3), np.int64), np.zeros((2048, 3), np.int64)
        for clearing in (0, 1):
            for from_origin in (0, 1):
                na = La.orc_cast_ray(o.ctypes.data_as(f32p), ta.ctypes.data_as(f32p), clearing, 1, 5.0, float(inv),
                                     float(4 / inv), from_origin, ba.ctypes.data_as(i64p), 2048)
                nb = Lb.orc_cast_ray(o.ctypes.data_as(f32p), ta.ctypes.data_as(f32p), clearing, 1, 5.0, float(inv),
                                     float(4 / inv), from_origin, bb.ctypes.data_as(i64p), 2048)
                assert na == nb and np.array_equal(ba[:min(na, 2048)], bb[:min(nb, 2048)])
        c1, c2 = int(rng.randint(0, 2 ** 32, dtype=np.uint64)), int(rng.randint(0, 2 ** 32, dtype=np.uint64))
        w1, w2 = float(np.float32(rng.uniform(0, 50))), float(np.float32(rng.uniform(1e-3, 5)))
        assert La.orc_blend_two_colors(c1, w1, c2, w2) == Lb.orc_blend_two_colors(c1, w1, c2, w2)
    for n in (1000, 5000, 307200):
        for s in (0, 1, 299, 300, 999, n // 2, n - 1):
            assert La.orc_mixed_index(s, n) == Lb.orc_mixed_index(s, n)
    offa, offb = np.zeros(78, np.int32), np.zeros(78, np.int32)
    da, db = np.zeros(26, np.float32), np.zeros(26, np.float32)
    La.orc_neighbor_lut(offa.ctypes.data_as(i32p), da.ctypes.data_as(f32p))
    Lb.orc_neighbor_lut(offb.ctypes.data_as(i32p), db.ctypes.data_as(f32p))
    assert np.array_equal(offa, offb) and np.array_equal(da.view(np.uint32), db.view(np.uint32))


def _same_mesh(a, b):
    assert set(a.keys()) == set(b.keys())
    nverts = 0
    for k in a:
        x, y = a[k], b[k]
        assert x["updated"] == y["updated"], k
        for f in ("vertices", "normals", "colors", "indices"):
            assert x[f].shape == y[f].shape, (k, f, x[f].shape, y[f].shape)
            assert np.array_equal(x[f].view(np.uint8), y[f].view(np.uint8)), (k, f)
        nverts += x["vertices"].shape[0]
    return nverts


@pytest.mark.parametrize("cfg", [dict(), dict(use_color=False), dict(min_weight=0.05)])
def test_mesh_integrator_bit_identical(cfg):
    """MeshIntegrator<TsdfVoxel>::generateMesh (mesh_integrator.h:142-392): incremental meshing of
    the kMesh-flagged blocks after every frame, then a full re-mesh; vertices, normals, colours,
    indices, `updated` and the cleared kMesh bits identical to the reference build."""
    out = []
    for L in (O.lib(), O.ref_lib()):
        L.orc_fast_reset_counter_set(0)
        m = O.OracleMap(0.1, 16, L=L)
        c = O.TsdfCfg()
        L.orc_tsdf_cfg_default(C.byref(c))
        c.default_truncation_distance = 0.4
        c.integrator_threads = 1
        it = m.tsdf_integrator("merged", c)
        ml = m.mesh_layer()
        snaps = []
        for k, (pose, pts, col) in enumerate(_frames(3)):
            it.integrate(pose[0], pose[1], pts, col)
            ml.generate(True, True, **cfg)
            snaps.append(ml.as_dict())
            if k == 1:
                ml.clear_updated()
        flags = {tuple(b): m.tsdf_block(b)[3] for b in m.block_indices()}
        ml.generate(False, False, **cfg)
        snaps.append(ml.as_dict())
        out.append((snaps, flags))
    (sa, fa), (sb, fb) = out
    assert fa == fb and all((v & 2) == 0 for v in fa.values())
    total = 0
    for x, y in zip(sa, sb):
        total += _same_mesh(x, y)
    assert total > 3000


@pytest.mark.parametrize("kind", ["simple", "merged", "fast"])
def test_axis_parallel_rays_quirk_q4_bit_identical(kind):
    """SURVEY Q4 (integrator_utils.cc:160-178): exactly axis-parallel rays divide by zero inside
    setupRayCaster; the walk that results (NaN / inf t-values through minCoeff) is the same in the
    restatement and in the reference's own ray caster."""
    q = np.array([1, 0, 0, 0], np.float32)
    dirs = np.array([[0, 0, 2.0], [0, 0, -1.5], [1.7, 0, 0], [-2.2, 0, 0], [0, 1.3, 0], [0, -0.9, 0],
                     [1.2, 1.2, 0], [0, -1.1, 2.3], [1.9, 0, -0.7], [0.4, 0.3, 2.0]], np.float32)
    col = np.full((dirs.shape[0], 4), 200, np.uint8)
    for pos in ([0.05, 0.05, 0.05], [0.0, 0.0, 0.0], [0.1, 0.25, -0.3], [-0.35, 0.2, 0.15]):
        frames = [((np.array(pos, np.float32), q), dirs, col)] * 2
        a, b = _both(kind, frames, voxel=0.1)
        _same_tsdf(a, b)
        assert a.num_blocks() > 0
