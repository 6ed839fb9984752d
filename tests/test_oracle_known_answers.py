"""Pins the oracle (CPU restatement) against every known-answer test the reference holds
for this path:
  test/test_tsdf_map.cc:36-360      index math incl. linear 8 <-> (0,1,0), 371 <-> (3,6,5),
                                    511 <-> (7,7,7), negative blocks, 101^3 origin round trip
  test/test_approx_hash_array.cc    ApproxHashSet insert / re-insert / recover rates, 50 resets
  test/test_bucket_queue.cc         BucketQueue pop order within 2 bucket widths
plus the observable constants SURVEY.md probed (hash wrap, derived floats, LUT order).
CPU only (no GPU).
"""
import ctypes as C

import numpy as np
import pytest


def _i32(*v):
    return np.array(v, np.int32)


def _i64(*v):
    return np.array(v, np.int64)


def _f32(*v):
    return np.array(v, np.float32)


def _pi32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _pi64(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _pf32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Grid:
    """voxel 0.1 / vps 8 fixture of TsdfMapTest (test_tsdf_map.cc:10-16)."""

    def __init__(self, oracle):
        self.L = oracle.lib()
        self.voxel = np.float32(0.1)
        self.vps = 8
        self.block_size = np.float32(self.voxel * np.float32(self.vps))
        self.voxel_inv = np.float32(1.0 / np.float64(self.voxel))
        self.block_inv = np.float32(1.0 / np.float64(self.block_size))

    def block_index(self, p):  # Layer::computeBlockIndexFromCoordinates, layer.h:127-131
        out = np.zeros(3, np.int64)
        self.L.orc_grid_index_from_point(_pf32(_f32(*p)), float(self.block_inv), _pi64(out))
        return tuple(int(v) for v in out)

    def block_origin(self, b):  # getOriginPointFromGridIndex, common.h:195-201
        out = np.zeros(3, np.float32)
        self.L.orc_origin_point_from_grid_index(_pi32(_i32(*b)), float(self.block_size), _pf32(out))
        return out

    def voxel_index(self, p, b):  # Block::computeTruncatedVoxelIndexFromCoordinates, block_inl.h:29-40
        rel = (_f32(*p) - self.block_origin(b)).astype(np.float32)
        out = np.zeros(3, np.int64)
        self.L.orc_grid_index_from_point(_pf32(rel), float(self.voxel_inv), _pi64(out))
        return tuple(int(min(max(v, 0), self.vps - 1)) for v in out)

    def linear(self, v):
        return int(self.L.orc_linear_index(_pi32(_i32(*v)), self.vps))

    def from_linear(self, lin):
        out = np.zeros(3, np.int32)
        self.L.orc_voxel_index_from_linear(lin, self.vps, _pi32(out))
        return tuple(int(x) for x in out)

    def center(self, v, b):  # Block::computeCoordinatesFromVoxelIndex, block.h:90-92
        out = np.zeros(3, np.float32)
        self.L.orc_center_point_from_grid_index(_pi64(_i64(*v)), float(self.voxel), _pf32(out))
        return self.block_origin(b) + out


def test_block_index_lookups(oracle):
    """test_tsdf_map.cc:36-132."""
    g = Grid(oracle)
    bs, vs = float(g.block_size), float(g.voxel)
    assert g.block_index((0, 0, 0)) == (0, 0, 0)
    assert g.block_index((0, vs, 0)) == (0, 0, 0)
    assert g.block_index((bs, bs, bs)) == (1, 1, 1)
    assert g.block_index((bs, bs + vs, bs)) == (1, 1, 1)
    assert g.block_index((-bs, -bs, -bs)) == (-1, -1, -1)
    assert g.block_index((-bs, -bs + vs, -bs)) == (-1, -1, -1)
    assert np.allclose(g.block_origin((1, 1, 1)), [bs] * 3, atol=1e-12)
    assert np.allclose(g.block_origin((-1, -1, -1)), [-bs] * 3, atol=1e-12)
    # Block::block_index() = round(origin * block_size_inv), block.h:157-159
    for b in ((0, 0, 0), (1, 1, 1), (-1, -1, -1), (-1, -1, 0)):
        out = np.zeros(3, np.int32)
        g.L.orc_grid_index_from_origin_point(_pf32(g.block_origin(b)), float(g.block_inv), _pi32(out))
        assert tuple(int(v) for v in out) == b
    # block allocation by coordinates (test_tsdf_map.cc:24-34): same block for 0.15 and 0.13
    assert g.block_index((0.0, 0.15, 0.0)) == g.block_index((0.0, 0.13, 0.0))
    assert g.block_index((-10.0, 13.5, 20.0)) != g.block_index((0.0, 0.15, 0.0))


@pytest.mark.parametrize("point,block,lin,vox", [
    ((0.0, 0.1, 0.0), (0, 0, 0), 8, (0, 1, 0)),            # test_tsdf_map.cc:137-160
    ((0.0, 0.0, 0.0), (0, 0, 0), 0, (0, 0, 0)),            # :161-184
    ((0.7, 0.7, 0.7), (0, 0, 0), 511, (7, 7, 7)),          # :185-213
    ((-0.8, -0.8, -0.8), (-1, -1, -1), 0, (0, 0, 0)),      # :215-244
    ((-1e-12, -1e-12, -1e-12), (-1, -1, -1), 511, (7, 7, 7)),  # :245-276
    ((-0.5, -0.2, 0.5), (-1, -1, 0), 371, (3, 6, 5)),      # :279-322
])
def test_voxel_index_known_answers(oracle, point, block, lin, vox):
    g = Grid(oracle)
    p = tuple(np.float32(np.float32(c)) for c in point)
    if block == (0, 0, 0) and lin == 511:  # 7.0 * voxel_size evaluated in double then float
        p = tuple(np.float32(7.0 * np.float64(np.float32(0.1))) for _ in range(3))
    if block == (-1, -1, -1) and lin == 0:
        p = tuple(-g.block_size for _ in range(3))
    if block == (-1, -1, 0):
        vs = np.float64(np.float32(0.1))
        p = (np.float32(-5.0 * vs), np.float32(-2.0 * vs), np.float32(5.0 * vs))
    if lin != 511 or block != (-1, -1, -1):  # the reference asserts no block lookup for (-eps)^3 (:245-276)
        assert g.block_index(p) == block
    v = g.voxel_index(p, block)
    assert v == vox
    assert g.linear(v) == lin
    assert g.from_linear(lin) == vox
    # centre of that voxel is within one voxel of the query point (EIGEN_MATRIX_NEAR(..., voxel_size))
    assert np.all(np.abs(g.center(vox, block) - np.array(p, np.float32)) <= g.voxel)


def test_block_origin_round_trip_101_cubed(oracle):
    """test_tsdf_map.cc:325-360: index -> origin -> index over [-50,50]^3 at block size 0.32."""
    L = oracle.lib()
    bs = np.float32(0.32)
    inv = np.float32(1.0 / np.float64(bs))   # constexpr FloatingPoint kBlockSizeInv = 1.0 / kBlockSize
    r = np.arange(-50, 51, dtype=np.int32)
    # vectorised restatement of the two one-liners, checked against the oracle on a sample
    origin = r.astype(np.float32) * bs
    back = np.round(origin * inv).astype(np.int32)
    assert np.array_equal(back, r)
    out_o = np.zeros(3, np.float32)
    out_i = np.zeros(3, np.int32)
    for x in (-50, -37, -1, 0, 1, 13, 50):
        for y in (-50, -2, 0, 29, 50):
            for z in (-50, 0, 7, 50):
                L.orc_origin_point_from_grid_index(_pi32(_i32(x, y, z)), float(bs), _pf32(out_o))
                L.orc_grid_index_from_origin_point(_pf32(out_o), float(inv), _pi32(out_i))
                assert tuple(out_i) == (x, y, z)


def test_global_local_block_round_trip_negative(oracle):
    """getBlockIndexFromGlobalVoxelIndex / getLocalFromGlobalVoxelIndex (common.h:215-243)."""
    L = oracle.lib()
    rng = np.random.RandomState(0)
    for vps in (8, 16, 32):
        for g in rng.randint(-5000, 5000, size=(200, 3)).astype(np.int64):
            b = np.zeros(3, np.int32); l = np.zeros(3, np.int32); back = np.zeros(3, np.int64)
            L.orc_block_index_from_global(_pi64(g), float(np.float32(1.0 / vps)), _pi32(b))
            L.orc_local_from_global(_pi64(g), vps, _pi32(l))
            assert np.all(l >= 0) and np.all(l < vps)
            assert np.array_equal(b, np.floor_divide(g, vps))
            L.orc_global_from_block_and_local(_pi32(b), _pi32(l), vps, _pi64(back))
            assert np.array_equal(back, g)


def test_hashes_wrap_and_truncate(oracle):
    """block_hash.h:20-31, 54-64: sum in size_t (mod 2^64), truncated to 32 bits."""
    L = oracle.lib()
    sl = 17191
    for idx in [(0, 0, 0), (1, 2, 3), (-1, -1, -1), (123456, -98765, 4242), (-7, 300000, -300000)]:
        want = (idx[0] + idx[1] * sl + idx[2] * sl * sl) % (1 << 64) % (1 << 32)
        assert L.orc_long_index_hash(_pi64(_i64(*idx))) == want
        assert L.orc_any_index_hash(_pi32(_i32(*idx))) == want
    assert L.orc_long_index_hash(_pi64(_i64(0, 0, 0))) == 0  # the hash-0 voxel of SURVEY Q6


def test_derived_float_constants(oracle):
    """1.0/0.05f -> exactly 20.0f; block_size(0.05f,16) = 0.800000012f."""
    assert np.float32(1.0 / np.float64(np.float32(0.05))) == np.float32(20.0)
    assert np.float32(np.float32(0.05) * np.float32(16)) == np.float32(0.800000012)


def test_mixed_thread_safe_index(oracle):
    """integrator_utils.cc:54-63: N=307200 -> 300 groups, stride-1024 interleave; tail identity."""
    L = oracle.lib()
    n = 307200
    assert [L.orc_mixed_index(s, n) for s in (0, 1, 2, 299, 300, 301)] == [0, 1024, 2048, 299 * 1024, 1, 1025]
    seen = np.array([L.orc_mixed_index(s, 5000) for s in range(5000)])
    assert np.array_equal(np.sort(seen), np.arange(5000))          # a permutation
    assert np.array_equal(seen[4096:], np.arange(4096, 5000))      # tail beyond groups*1024


def test_blend_two_colors(oracle):
    """common.h:105-125: per channel (uint8) round(c1*w1/(w1+w2) + c2*w2/(w1+w2))."""
    L = oracle.lib()

    def pack(r, g, b, a):
        return r | (g << 8) | (b << 16) | (a << 24)
    assert L.orc_blend_two_colors(pack(0, 0, 0, 0), 0.0, pack(10, 20, 30, 255), 1.0) == pack(10, 20, 30, 255)
    assert L.orc_blend_two_colors(pack(100, 0, 255, 255), 1.0, pack(200, 1, 0, 255), 1.0) == pack(150, 1, 128, 255)
    assert L.orc_blend_two_colors(pack(7, 7, 7, 7), 3.0, pack(8, 8, 8, 8), 1.0) == pack(7, 7, 7, 7)


def test_approx_hash_set_rates(oracle):
    """test_approx_hash_array.cc:59-109 with ApproxHashSet<16,10>: > 950/1000 inserted,
    < 50 re-inserted, over 50 resets; > 950 recovered at the end."""
    L = oracle.lib()
    rng = np.random.RandomState(1)
    idx = rng.randint(1, 2 ** 31 - 1, size=(1000, 3)).astype(np.int32)
    hashes = [L.orc_any_index_hash(_pi32(np.ascontiguousarray(i))) for i in idx]
    s = L.orc_approx_set_create(1)
    try:
        for _ in range(50):
            L.orc_approx_set_reset(s)
            assert sum(L.orc_approx_set_replace_hash(s, h) for h in hashes) > 950
            assert sum(L.orc_approx_set_replace_hash(s, h) for h in hashes) < 50
        assert sum(L.orc_approx_set_is_present(s, h) for h in hashes) > 950
    finally:
        L.orc_approx_set_destroy(s)


def test_approx_hash_set_hash0_quirk(oracle):
    """after the first resetApproxSet() (offset 1) the zero-initialised slot makes
    hash 0 read as already present; at offset 0 the max() sentinel prevents that."""
    L = oracle.lib()
    s = L.orc_approx_set_create(0)
    try:
        assert L.orc_approx_set_is_present(s, 0) == 0       # ctor sentinel at offset 0
        L.orc_approx_set_reset(s)
        assert L.orc_approx_set_is_present(s, 0) == 1       # the quirk
        assert L.orc_approx_set_replace_hash(s, 0) == 0
        assert L.orc_approx_set_replace_hash(s, 1 << 20) == 1  # same slot, other hash evicts
        assert L.orc_approx_set_is_present(s, 0) == 0
    finally:
        L.orc_approx_set_destroy(s)


def test_bucket_queue_order(oracle):
    """test_bucket_queue.cc:12-66: pops are monotone in |value| within two bucket widths."""
    L = oracle.lib()
    rng = np.random.RandomState(0)
    n, max_d, nb = 100, 20.0, 20
    q = L.orc_bucket_queue_create(nb, max_d)
    try:
        dist = rng.uniform(-max_d, max_d, n)
        max_diff = 2 * max_d / (nb - 1)

        def drain(count=None):
            last = 0.0
            k = 0
            while not L.orc_bucket_queue_empty(q) and (count is None or k < count):
                v = dist[L.orc_bucket_queue_front(q)]
                L.orc_bucket_queue_pop(q)
                assert abs(abs(v) - abs(last)) < max_diff
                last = v
                k += 1
        for i in range(n):
            L.orc_bucket_queue_push(q, i, float(dist[i]))
        drain()
        for i in range(n):
            L.orc_bucket_queue_push(q, i, float(dist[i]))
        drain(n // 2)
        for i in range(n // 2):
            L.orc_bucket_queue_push(q, i, float(dist[i]))
        drain()
    finally:
        L.orc_bucket_queue_destroy(q)


def test_neighbor_lut_order(oracle):
    """neighbor_tools.cc:8-34: 6 faces, 12 edges, 8 corners in the reference's column order."""
    L = oracle.lib()
    off = np.zeros(78, np.int32); dist = np.zeros(26, np.float32)
    L.orc_neighbor_lut(_pi32(off), _pf32(dist))
    off = off.reshape(26, 3)
    assert [tuple(o) for o in off[:6]] == [(-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1)]
    assert tuple(off[6]) == (-1, -1, 0) and tuple(off[17]) == (1, 0, 1)
    assert tuple(off[18]) == (-1, -1, -1) and tuple(off[25]) == (1, 1, 1)
    assert len({tuple(o) for o in off}) == 26
    want = np.sqrt((off.astype(np.float64) ** 2).sum(1)).astype(np.float32)
    assert np.array_equal(dist, want)


def test_raycaster_emits_manhattan_plus_one(oracle):
    """integrator_utils.cc:111-179: |dx|+|dy|+|dz|+1 indices, 6-connected, first = start voxel,
    last = end voxel for a generic ray; Fast casts the same voxels in reverse order."""
    L = oracle.lib()
    o = _f32(0.13, -0.21, 0.07); p = _f32(2.37, 1.11, -0.93)
    buf = np.zeros((4096, 3), np.int64)
    n = L.orc_cast_ray(_pf32(o), _pf32(p), 0, 1, 5.0, float(np.float32(20.0)), 0.2, 1, _pi64(buf), 4096)
    fwd = buf[:n].copy()
    n2 = L.orc_cast_ray(_pf32(o), _pf32(p), 0, 1, 5.0, float(np.float32(20.0)), 0.2, 0, _pi64(buf), 4096)
    rev = buf[:n2].copy()
    assert n == np.abs(fwd[-1] - fwd[0]).sum() + 1
    assert np.all(np.abs(np.diff(fwd, axis=0)).sum(1) == 1)
    assert tuple(fwd[0]) == (2, -5, 1)   # floor(origin * 20 + 1e-6)
    assert n2 == n and set(map(tuple, rev)) >= {tuple(fwd[0]), tuple(fwd[-1])}
    assert tuple(rev[0]) == tuple(fwd[-1]) and tuple(rev[-1]) == tuple(fwd[0])


def test_weight_chain_integer_prefix_model(tmp_path):
    """The claim behind the fold's parallel weight chain (weight_stretches in vbx_kernels_tsdf.hpp): inside
    one binade the float chain W <- min(max_weight, W + w) is an integer prefix sum.  A C model of the same
    stretch procedure is compared bit for bit with the sequential loop on 400 k random chunks."""
    import os, subprocess
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "weight_chain_model.c")
    exe = str(tmp_path / "weight_chain_model")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, src, "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("ok chunks 400000")
