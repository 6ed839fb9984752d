"""-m gpu: HIP path vs the CPU oracle on the same seeded inputs, through the C-ABI.

Bar: voxel indices (allocated blocks + observed-voxel masks) bit-exact; distances/weights
are compared BIT-exact as well (stronger than north_star's 1e-4) because the per-voxel fold
replays the reference's 1-thread visiting order; colours bit-exact.
"""
import numpy as np
import pytest

from parity_utils import compare_tsdf, layer_stats
from voxblox_amd import scenes

pytestmark = pytest.mark.gpu


def _cfgs(oracle, trunc, **kw):
    from voxblox_amd import capi
    okw = {k: v for k, v in kw.items() if k not in ("merged_bundle_order", "fast_observed_set")}   # HIP-only fields
    gkw = {k: v for k, v in kw.items() if not k.startswith("oracle_")}
    if kw.get("oracle_merged_sorted_bundles"):
        gkw["merged_bundle_order"] = 1      # ascending voxel key on both sides
    if kw.get("oracle_fast_exact_observed_set"):
        gkw["fast_observed_set"] = 1        # exact observed-voxel set on both sides
    return (oracle.tsdf_cfg(default_truncation_distance=trunc, integrator_threads=1, **okw),
            capi.tsdf_cfg(default_truncation_distance=trunc, **gkw))


def _run(oracle, kind, voxel, frames, max_blocks=4096, **kw):
    from voxblox_amd import capi
    ocfg, gcfg = _cfgs(oracle, 4 * voxel, **kw)
    oracle.lib().orc_fast_reset_counter_set(0)
    capi.lib().vbx_fast_reset_counter_set(0)   # the library's process-wide counter (tsdf_integrator.cc:564), like the oracle's
    om = oracle.OracleMap(voxel, 16)
    oi = om.tsdf_integrator(kind, ocfg)
    gm = capi.Map(voxel, 16, max_blocks=max_blocks)
    k = {"simple": capi.TSDF_SIMPLE, "merged": capi.TSDF_MERGED, "fast": capi.TSDF_FAST}[kind]
    gm.cum = {}   # counters summed over the frames (the oracle's stats are cumulative too)
    for pose, pts, col in frames:
        oi.integrate(pose[0], pose[1], pts, col)
        gm.integrate(k, gcfg, pose[0], pose[1], pts, col)
        for name, v in gm.counters().items():
            gm.cum[name] = gm.cum.get(name, 0) + v
    return om, oi, gm


def _small_room(k, n=100):
    return scenes.room_frame(k, n, f=80.0, width=160, height=120)


def test_simple_plane_config1(oracle):
    """BASELINE config 1: full 640x480 plane frame, 0.10 m voxels."""
    frames = [scenes.plane_frame()]
    om, oi, gm = _run(oracle, "simple", 0.10, frames)
    st = compare_tsdf(gm.tsdf_dict(), om.tsdf_dict(), exact=True)
    c = gm.counters()
    assert c["voxel_updates"] == oi.stats()["voxel_updates"]
    # every voxel a ray visits is touched; voxels whose summed weight stays below 1e-6 are visited but
    # never observed (updateTsdfVoxel returns early, tsdf_integrator.cc:192-194)
    assert c["voxels_touched"] >= st["observed_voxels"] > 0
    assert c["rays_cast"] == oi.stats()["rays_cast"]


def test_simple_room_stream(oracle):
    frames = [_small_room(k) for k in (0, 7, 14)]
    om, oi, gm = _run(oracle, "simple", 0.05, frames)
    compare_tsdf(gm.tsdf_dict(), om.tsdf_dict(), exact=True)
    assert gm.counters()["points"] == frames[-1][1].shape[0]


def test_simple_no_carving_const_weight(oracle):
    frames = [_small_room(3)]
    om, oi, gm = _run(oracle, "simple", 0.05, frames, voxel_carving_enabled=0, use_const_weight=1,
                      use_weight_dropoff=0)
    compare_tsdf(gm.tsdf_dict(), om.tsdf_dict(), exact=True)


def test_merged_room_stream(oracle):
    frames = [_small_room(k) for k in (0, 5, 10)]
    om, oi, gm = _run(oracle, "merged", 0.05, frames, oracle_merged_sorted_bundles=1)
    compare_tsdf(gm.tsdf_dict(), om.tsdf_dict(), exact=True)
    assert gm.counters()["rays_cast"] == oi.stats()["bundles"] + oi.stats()["clear_bundles"]


def test_merged_anti_grazing(oracle):
    frames = [_small_room(2)]
    om, oi, gm = _run(oracle, "merged", 0.10, frames, oracle_merged_sorted_bundles=1,
                      enable_anti_grazing=1)
    compare_tsdf(gm.tsdf_dict(), om.tsdf_dict(), exact=True)


@pytest.mark.parametrize("extra", [{}, {"enable_anti_grazing": 1}, {"integration_order_mode": 1}])
def test_merged_reference_bundle_order_bit_exact(oracle, extra):
    """Default merged_bundle_order = 0: the bundles are visited in the iteration order of the
    reference's std::unordered_map (replayed on the host with the same container and hash), so
    the HIP result equals the UNSWITCHED 1-thread reference bit for bit."""
    frames = [_small_room(k) for k in (0, 5, 10)] if not extra.get("integration_order_mode") \
        else [_unique_norm_frame(k) for k in (0, 5)]
    om, oi, gm = _run(oracle, "merged", 0.05, frames, **extra)   # oracle with the reference's own order
    compare_tsdf(gm.tsdf_dict(), om.tsdf_dict(), exact=True)
    assert gm.counters()["rays_cast"] == oi.stats()["bundles"] + oi.stats()["clear_bundles"]


def test_merged_sorted_bundle_order_vs_reference_envelope(oracle):
    """merged_bundle_order = 1 (ascending voxel key, no host step) against the reference's order:
    same voxels observed; distances differ only where the clamped fold is order-sensitive —
    bounded by the reference's own envelope (test_sdf_integrators.cc:162-178)."""
    frames = [_small_room(k) for k in (0, 5)]
    om, oi, gm = _run(oracle, "merged", 0.05, frames, merged_bundle_order=1)  # oracle in reference order
    g, r = gm.tsdf_dict(), om.tsdf_dict()
    assert set(g.keys()) == set(r.keys())
    st = layer_stats(g, r)
    assert st["a_only"] == 0 and st["b_only"] == 0
    assert st["rmse"] < 2 * 0.05


def test_fast_room_stream_exact_observed_set(oracle):
    frames = [_small_room(k) for k in (0, 4, 8)]
    om, oi, gm = _run(oracle, "fast", 0.05, frames, oracle_fast_exact_observed_set=1)
    compare_tsdf(gm.tsdf_dict(), om.tsdf_dict(), exact=True)
    assert gm.cum["rays_cast"] == oi.stats()["rays_cast"]
    assert gm.cum["voxel_updates"] == oi.stats()["voxel_updates"]


@pytest.mark.parametrize("extra", [{}, {"max_consecutive_ray_collisions": 0}, {"clear_checks_every_n_frames": 3},
                                   {"start_voxel_subsampling_factor": 1.0, "max_consecutive_ray_collisions": 5}])
def test_fast_reference_observed_set_bit_exact(oracle, extra):
    """Default fast_observed_set = 0: voxel_observed_approx_set_ is the reference's lossy
    2^20-slot ApproxHashSet, replayed exactly (evictions, offset resets, persistence over
    clear_checks_every_n_frames) — the HIP result equals the UNSWITCHED 1-thread reference bit
    for bit."""
    frames = [_small_room(k) for k in (0, 4, 8, 12)]
    om, oi, gm = _run(oracle, "fast", 0.05, frames, **extra)      # oracle = true reference semantics
    compare_tsdf(gm.tsdf_dict(), om.tsdf_dict(), exact=True)


def test_fast_approx_set_offset_wrap(oracle):
    """resetApproxSet's full reset (approx_hash_array.h:156-169): 9997 empty clouds with
    clear_checks_every_n_frames = 1 move both ApproxHashSets to offset 9997 (an empty cloud still
    counts as a frame, tsdf_integrator.cc:564-569); the real frames then run at offsets 9998,
    9999, 0 (sets zeroed, sentinel back in slot 0) and 1."""
    empty = (np.zeros((0, 3), np.float32), np.zeros((0, 4), np.uint8))
    real = [_small_room(k) for k in (0, 4, 8, 12)]
    frames = [(real[0][0], empty[0], empty[1])] * 9997 + real
    om, oi, gm = _run(oracle, "fast", 0.05, frames, clear_checks_every_n_frames=1)
    compare_tsdf(gm.tsdf_dict(), om.tsdf_dict(), exact=True)


def test_fast_empty_clouds_count_as_frames(oracle):
    """clear_checks_every_n_frames = 3 with empty clouds interleaved: the reset counter advances
    on every integratePointCloud call, so the sets are cleared at the same real frames as in the
    reference."""
    empty = (np.zeros((0, 3), np.float32), np.zeros((0, 4), np.uint8))
    real = [_small_room(k) for k in (0, 4, 8, 12)]
    e = (real[0][0], empty[0], empty[1])
    frames = [real[0], e, real[1], e, e, real[2], real[3]]
    om, oi, gm = _run(oracle, "fast", 0.05, frames, clear_checks_every_n_frames=3)
    compare_tsdf(gm.tsdf_dict(), om.tsdf_dict(), exact=True)


def test_fast_reference_observed_set_full_frame(oracle):
    """The same at BASELINE configs[1] size: two full 640x480 frames at 0.05 m."""
    frames = [scenes.room_frame(k, 100) for k in (0, 1)]
    om, oi, gm = _run(oracle, "fast", 0.05, frames, max_blocks=8192)
    compare_tsdf(gm.tsdf_dict(), om.tsdf_dict(), exact=True)


def test_fast_exact_set_vs_reference_approx_set_envelope(oracle):
    """fast_observed_set = 1 (exact set, one solve) against the reference's ApproxHashSet
    semantics: inside the reference's test envelope for Fast-vs-Simple
    (test_sdf_integrators.cc:162-178): same rays cast, overlap within 1 %, small rmse."""
    frames = [_small_room(k) for k in (0, 4)]
    om, oi, gm = _run(oracle, "fast", 0.05, frames, fast_observed_set=1)  # oracle: true reference semantics
    g, r = gm.tsdf_dict(), om.tsdf_dict()
    st = layer_stats(g, r)
    total = st["both"] + st["a_only"] + st["b_only"]
    assert (st["a_only"] + st["b_only"]) <= 0.01 * total, st
    assert st["rmse"] < 2 * 0.05, st


def _unique_norm_frame(k):
    """Frame whose points have pairwise distinct squared norms: "sorted" order is then unique
    (the reference's std::sort leaves ties unspecified, integrator_utils.cc:24-37)."""
    pose, pts, col = _small_room(k)
    sq = pts[:, 0] * pts[:, 0] + (pts[:, 1] * pts[:, 1] + pts[:, 2] * pts[:, 2])  # Eigen's order
    _, first = np.unique(sq.astype(np.float32), return_index=True)
    keep = np.sort(first)
    return pose, np.ascontiguousarray(pts[keep]), np.ascontiguousarray(col[keep])


@pytest.mark.parametrize("kind,extra", [("simple", {}), ("merged", {"oracle_merged_sorted_bundles": 1}),
                                        ("fast", {"oracle_fast_exact_observed_set": 1})])
def test_sorted_integration_order(oracle, kind, extra):
    """integration_order_mode = "sorted" (SortedThreadSafeIndex): nearest points first."""
    frames = [_unique_norm_frame(k) for k in (1, 6)]
    om, oi, gm = _run(oracle, kind, 0.05, frames, integration_order_mode=1, **extra)
    compare_tsdf(gm.tsdf_dict(), om.tsdf_dict(), exact=True)
    # and the order matters: mixed order gives a different layer for the order-sensitive fold
    if kind == "simple":
        _, _, gm2 = _run(oracle, kind, 0.05, frames)
        a, b = gm.tsdf_dict(), gm2.tsdf_dict()
        assert any(not np.array_equal(a[k][0], b[k][0]) for k in a)


@pytest.mark.parametrize("kind", ["simple", "merged", "fast"])
def test_sparsity_compensation_factor_on_the_device(oracle, kind):
    """use_sparsity_compensation_factor = 1 (tsdf_integrator.cc:173-182): the weight of every update whose |sdf| lies
    inside the truncation band is multiplied by the factor — all three integrators, bit for bit against the oracle
    (which test_oracle_vs_reference_build.py pins against the reference build with the same Config)."""
    frames = [_small_room(k) for k in (0, 6, 12)]
    om, oi, gm = _run(oracle, kind, 0.05, frames, use_sparsity_compensation_factor=1, sparsity_compensation_factor=3.0)
    st = compare_tsdf(gm.tsdf_dict(), om.tsdf_dict(), exact=True)
    assert st["observed_voxels"] > 10000
    # and the switch does something: without it the weights inside the band are a third
    _, _, g0 = _run(oracle, kind, 0.05, frames)
    a, b = gm.tsdf_dict(), g0.tsdf_dict()
    assert a.keys() == b.keys()
    assert any(not np.array_equal(a[k][1], b[k][1]) for k in a)


def test_fast_reset_counter_is_shared_by_all_integrators_of_the_process(oracle):
    """`static int64_t reset_counter` (tsdf_integrator.cc:564-569) counts the calls of EVERY FastTsdfIntegrator of the
    process: two integrators on two maps with clear_checks_every_n_frames = 3, called alternately, clear their sets when
    the SHARED counter reaches 3 — i.e. on calls 3, 6, 9 of the interleaved sequence, whichever integrator makes them.
    Per-handle counters would clear each map's sets on its own 3rd call instead: a different map."""
    from voxblox_amd import capi
    ocfg, gcfg = _cfgs(oracle, 0.2, clear_checks_every_n_frames=3)
    oracle.lib().orc_fast_reset_counter_set(0)
    capi.lib().vbx_fast_reset_counter_set(0)
    oms = [oracle.OracleMap(0.05, 16) for _ in range(2)]
    ois = [om.tsdf_integrator("fast", ocfg) for om in oms]
    gms = [capi.Map(0.05, 16, max_blocks=4096) for _ in range(2)]
    # (poses close together so that what a set still holds decides which rays are cast)
    seq = [(0, 0), (1, 0), (0, 1), (1, 1), (0, 2), (1, 2), (0, 3), (1, 3)]
    for which, k in seq:
        pose, pts, col = _small_room(k)
        ois[which].integrate(pose[0], pose[1], pts, col)
        gms[which].integrate(capi.TSDF_FAST, gcfg, pose[0], pose[1], pts, col)
    assert capi.lib().vbx_fast_reset_counter_get() == len(seq) % 3
    for om, gm in zip(oms, gms):
        compare_tsdf(gm.tsdf_dict(), om.tsdf_dict(), exact=True)
    # the same calls with a counter per map (each map alone in the process) give another layer for at least one of the two
    differs = False
    for which in range(2):
        oracle.lib().orc_fast_reset_counter_set(0)
        om1 = oracle.OracleMap(0.05, 16)
        oi1 = om1.tsdf_integrator("fast", ocfg)
        for w, k in seq:
            if w == which:
                pose, pts, col = _small_room(k)
                oi1.integrate(pose[0], pose[1], pts, col)
        a, b = om1.tsdf_dict(), oms[which].tsdf_dict()
        differs |= a.keys() != b.keys() or any(not np.array_equal(a[k][0], b[k][0]) or not np.array_equal(a[k][1], b[k][1]) for k in a)
    assert differs, "the scenario does not tell a shared counter from per-map counters"


@pytest.mark.parametrize("n_frames", [2, 3])
def test_fast_clear_checks_every_n_frames(oracle, n_frames):
    """clear_checks_every_n_frames > 1: both voxel sets survive n frames (tsdf_integrator.cc:564-569)."""
    frames = [_small_room(k) for k in (0, 1, 2, 3, 4)]
    om, oi, gm = _run(oracle, "fast", 0.05, frames, oracle_fast_exact_observed_set=1,
                      clear_checks_every_n_frames=n_frames)
    compare_tsdf(gm.tsdf_dict(), om.tsdf_dict(), exact=True)
    # fewer voxels are re-observed than with a per-frame reset
    _, _, gm1 = _run(oracle, "fast", 0.05, frames, oracle_fast_exact_observed_set=1)
    assert gm.tsdf_dict().keys() <= gm1.tsdf_dict().keys()


def test_fast_and_simple_fine_voxels_0p02(oracle):
    """BASELINE configs[4] resolution (0.02 m, truncation 0.08): per-ray lists of several hundred
    voxels, ~10x the blocks; bit-exact like the 0.05 m cases."""
    frames = [_small_room(k) for k in (0, 9)]
    om, oi, gm = _run(oracle, "fast", 0.02, frames, max_blocks=32768, oracle_fast_exact_observed_set=1)
    compare_tsdf(gm.tsdf_dict(), om.tsdf_dict(), exact=True)
    assert gm.num_blocks() > 2000
    om, oi, gm = _run(oracle, "simple", 0.02, frames[:1], max_blocks=32768)
    compare_tsdf(gm.tsdf_dict(), om.tsdf_dict(), exact=True)


def test_fast_reference_set_fine_voxels_full_resolution(oracle):
    """BASELINE configs[4]'s resolution with the DEFAULT observed-voxel set (the reference's ApproxHashSet,
    tsdf_integrator.cc:531-551, approx_hash_array.h:125-134): three consecutive full 640x480 room frames
    at 0.02 m.  ~23 M probes per frame land in the set's 2^20 slots, so voxels evict each other all the
    time, rays run several times further than under an exact set and the replay leaves its whole-frame
    rounds for blocks of consecutive rays (vbx_host_tsdf.hpp, "Phase 2") — the counters prove that path
    ran; the map must still equal the 1-thread reference bit for bit."""
    frames = [scenes.room_frame(k, 100) for k in (0, 1, 2)]
    om, oi, gm = _run(oracle, "fast", 0.02, frames, max_blocks=131072)
    compare_tsdf(gm.tsdf_dict(), om.tsdf_dict(), exact=True)
    assert gm.num_blocks() > 2000
    assert gm.cum["rays_cast"] == oi.stats()["rays_cast"]
    assert gm.cum["voxel_updates"] == oi.stats()["voxel_updates"] > 50_000_000
    assert gm.cum["replay_rounds"] >= 3 * 10          # long dependency chains: tens of rounds per frame
    assert gm.cum["replay_block_rounds"] >= 3 * 16    # ... most of them on blocks of consecutive rays (Phase 2)


@pytest.mark.parametrize("kind", ["simple", "merged", "fast"])
def test_edge_case_clouds(oracle, kind):
    """Degenerate inputs the reference handles silently: an empty cloud, a cloud whose points are all
    rejected by isPointValid (tsdf_integrator.h:112-129), a single point, only-clearing rays (beyond
    max_ray_length_m), freespace points, duplicates of one point, and a frame after them that must
    integrate as if nothing had happened."""
    from voxblox_amd import capi
    voxel = 0.1
    ocfg, gcfg = _cfgs(oracle, 4 * voxel)
    oracle.lib().orc_fast_reset_counter_set(0)
    om = oracle.OracleMap(voxel, 16)
    oi = om.tsdf_integrator(kind, ocfg)
    gm = capi.Map(voxel, 16, max_blocks=2048)
    k = {"simple": capi.TSDF_SIMPLE, "merged": capi.TSDF_MERGED, "fast": capi.TSDF_FAST}[kind]
    pos = np.array([0.1, 0.2, 0.3], np.float32)
    q = np.array([1, 0, 0, 0], np.float32)
    white = lambda n: np.full((n, 4), 255, np.uint8)
    clouds = [
        (np.zeros((0, 3), np.float32), False),                                    # empty
        (np.full((5, 3), 0.01, np.float32), False),                               # all closer than min_ray_length_m
        (np.array([[0.3, 0.1, 2.0]], np.float32), False),                         # one point
        (np.array([[0.5, 0.2, 7.0], [-1.0, 0.4, 9.0]], np.float32), False),       # clearing rays only
        (np.array([[0.2, -0.3, 1.5], [0.4, 0.3, 2.5]], np.float32), True),        # freespace_points = true
        (np.tile(np.array([[0.7, 0.1, 1.8]], np.float32), (300, 1)), False),      # 300 copies of one point
    ]
    for pts, freespace in clouds:
        col = white(pts.shape[0])
        oi.integrate(pos, q, pts, col, freespace)
        gm.integrate(k, gcfg, pos, q, pts, col, freespace)
        g, r = gm.tsdf_dict(), om.tsdf_dict()
        assert set(g) == set(r)
        if r:
            compare_tsdf(g, r, exact=True)
    pose, pts, col = _small_room(3)
    oi.integrate(pose[0], pose[1], pts, col)
    gm.integrate(k, gcfg, pose[0], pose[1], pts, col)
    compare_tsdf(gm.tsdf_dict(), om.tsdf_dict(), exact=True)


@pytest.mark.parametrize("kind", ["simple", "merged", "fast"])
def test_axis_parallel_rays_quirk_q4(oracle, kind):
    """`std::abs(x) < 0.0` never holds (integrator_utils.cc:160-178), so a ray component
    that is exactly 0 divides by zero: t_step = NaN, t_to_next = -inf / +inf / NaN, and Eigen's
    first-wins minCoeff decides what the walk does (frozen on the start voxel, or one duplicate
    emission and a ray that ends a voxel short).  Exactly axis-parallel rays from a sensor at a voxel
    centre / on a voxel face / on a voxel edge: the HIP ray caster replays the same IEEE arithmetic."""
    from voxblox_amd import capi
    voxel = 0.1
    ocfg, gcfg = _cfgs(oracle, 4 * voxel)
    k = {"simple": capi.TSDF_SIMPLE, "merged": capi.TSDF_MERGED, "fast": capi.TSDF_FAST}[kind]
    q = np.array([1, 0, 0, 0], np.float32)
    dirs = np.array([[0, 0, 2.0], [0, 0, -1.5], [1.7, 0, 0], [-2.2, 0, 0], [0, 1.3, 0], [0, -0.9, 0],   # one non-zero
                     [1.2, 1.2, 0], [0, -1.1, 2.3], [1.9, 0, -0.7],                                     # two non-zero
                     [0.4, 0.3, 2.0]], np.float32)                                                      # generic
    for pos in ([0.05, 0.05, 0.05], [0.0, 0.0, 0.0], [0.1, 0.25, -0.3], [-0.35, 0.2, 0.15]):
        oracle.lib().orc_fast_reset_counter_set(0)
        om = oracle.OracleMap(voxel, 16)
        oi = om.tsdf_integrator(kind, ocfg)
        gm = capi.Map(voxel, 16, max_blocks=2048)
        pos = np.array(pos, np.float32)
        col = np.full((dirs.shape[0], 4), 200, np.uint8)
        for rep in range(2):
            oi.stats(reset=True)
            oi.integrate(pos, q, dirs, col)
            gm.integrate(k, gcfg, pos, q, dirs, col)
        g, r = gm.tsdf_dict(), om.tsdf_dict()
        assert set(g) == set(r), (pos, sorted(set(g) ^ set(r)))
        compare_tsdf(g, r, exact=True)
        if kind != "fast":   # the frozen / shortened walks visit the same number of voxels
            assert gm.counters()["voxel_updates"] == oi.stats()["voxel_updates"]


@pytest.mark.parametrize("kind", ["simple", "merged", "fast"])
def test_non_finite_points_are_dropped(oracle, kind):
    """NaN / inf points reach the reference only when a caller skips the finite filter of
    conversions.h:135-137, and then read uninitialised RayCaster members; oracle and HIP path both
    treat them as dropped points.  The finite points around them integrate as usual."""
    from voxblox_amd import capi
    voxel = 0.1
    ocfg, gcfg = _cfgs(oracle, 4 * voxel)
    k = {"simple": capi.TSDF_SIMPLE, "merged": capi.TSDF_MERGED, "fast": capi.TSDF_FAST}[kind]
    pose, pts, col = _small_room(2)
    pts = pts.copy()
    pts[5] = [np.nan, 0.1, 1.0]
    pts[77] = [0.3, np.inf, 2.0]
    pts[300] = [np.nan, np.nan, np.nan]
    pts[1000] = [0.2, 0.1, -np.inf]
    oracle.lib().orc_fast_reset_counter_set(0)
    om = oracle.OracleMap(voxel, 16)
    oi = om.tsdf_integrator(kind, ocfg)
    gm = capi.Map(voxel, 16, max_blocks=2048)
    oi.integrate(pose[0], pose[1], pts, col)
    gm.integrate(k, gcfg, pose[0], pose[1], pts, col)
    compare_tsdf(gm.tsdf_dict(), om.tsdf_dict(), exact=True)


@pytest.mark.parametrize("kind", ["simple", "merged", "fast"])
def test_failures_are_loud(kind):
    """No silent degradation: a block pool that may not grow any further (vbx_set_pool_limit) and coordinates
    outside the key range fail the call with an error the caller can read."""
    from voxblox_amd import capi
    k = {"simple": capi.TSDF_SIMPLE, "merged": capi.TSDF_MERGED, "fast": capi.TSDF_FAST}[kind]
    cfg = capi.tsdf_cfg(default_truncation_distance=0.4)
    pose, pts, col = _small_room(0)
    gm = capi.Map(0.1, 16, max_blocks=8)
    gm.set_pool_limit(16)
    with pytest.raises(capi.VbxError, match="capacity"):
        gm.integrate(k, cfg, pose[0], pose[1], pts, col)
    gm = capi.Map(0.1, 16, max_blocks=1024)
    far = np.array([120000.0, 0.0, 0.0], np.float32)      # 1.2e6 voxels from the origin
    with pytest.raises(capi.VbxError, match="2\\^20 voxels"):
        gm.integrate(k, cfg, far, pose[1], pts, col)
    assert gm.num_blocks() == 0                           # nothing was integrated from the rejected cloud
    gm.integrate(k, cfg, pose[0], pose[1], pts, col)      # and the map is still usable
    assert gm.num_blocks() > 10


@pytest.mark.parametrize("kind", ["simple", "merged", "fast"])
def test_block_pool_grows_on_demand(oracle, kind):
    """Layer::allocateBlockPtrByIndex never fails (layer.h:133-160): a map created with room for 8 blocks
    integrates a stream that needs ~80 — the pool doubles as often as it takes (new arrays, used slots
    copied, hash table rebuilt, the allocation pass of the running call repeated) and the result equals the
    oracle's bit for bit, exactly as with a pool that was large from the start."""
    frames = [_small_room(k) for k in (0, 6, 12)]
    om, oi, gm = _run(oracle, kind, 0.1, frames, max_blocks=8)
    compare_tsdf(gm.tsdf_dict(), om.tsdf_dict(), exact=True)
    assert gm.num_blocks() > 40


@pytest.mark.parametrize("kind,extra", [("simple", {}), ("simple", {"use_const_weight": 1}),
                                        ("simple", {"max_weight": 300.0}), ("simple", {"use_weight_dropoff": 0}),
                                        ("merged", {"use_const_weight": 1, "max_weight": 2000.0})])
def test_giant_runs_full_resolution_stream(oracle, kind, extra):
    """640x480 frames from nearly the same pose: the sensor's own voxel collects one update per ray (300k
    in a row), its neighbours tens of thousands.  Those runs are folded by a workgroup in rounds of 4096
    updates under the claim 'distance and colour stay, the weight advances by integer steps inside its
    binade' (fold_giant_runs) with every update verified literally; the weight passes through all binades up
    to max_weight in the first frame, saturates, and is then an identity — all bit-exact against the
    1-thread order."""
    frames = [scenes.room_frame(k, 100) for k in (0, 1, 2, 3)]
    om, oi, gm = _run(oracle, kind, 0.05, frames, max_blocks=8192, **extra)
    st = compare_tsdf(gm.tsdf_dict(), om.tsdf_dict(), exact=True)
    assert st["observed_voxels"] > 100000
    w = np.concatenate([np.asarray(v[1]).ravel() for v in gm.tsdf_dict().values()])
    assert (w == np.float32(extra.get("max_weight", 10000.0))).sum() > 10      # saturated voxels exist


@pytest.mark.parametrize("kind,scene,n_frames", [("fast", "room", 40), ("merged", "cow", 16)])
def test_long_full_resolution_streams_bit_exact(oracle, kind, scene, n_frames):
    """Soak at BASELINE size: 40 consecutive 640x480 frames of the configs[1] room stream through the
    Fast integrator (reference observed-voxel set), 16 of the configs[2] orbit through Merged
    (reference bundle order) — the map after the last frame equals the oracle's bit for bit, so the
    rare paths (list rebuilds after block allocation, long replay chains, rehash boundaries of the
    bundle map) are exact too, not only the first frames."""
    import hashlib
    from voxblox_amd import capi
    voxel = 0.05
    ocfg, gcfg = _cfgs(oracle, 4 * voxel)
    oracle.lib().orc_fast_reset_counter_set(0)
    om = oracle.OracleMap(voxel, 16)
    oi = om.tsdf_integrator(kind, ocfg)
    gm = capi.Map(voxel, 16, max_blocks=16384)
    k = {"merged": capi.TSDF_MERGED, "fast": capi.TSDF_FAST}[kind]
    for i in range(n_frames):
        pose, pts, col = scenes.room_frame(i, 100) if scene == "room" else scenes.cow_and_lady_like_frame(5 * i)
        oi.integrate(pose[0], pose[1], pts, col)
        gm.integrate(k, gcfg, pose[0], pose[1], pts, col)
    g, r = gm.tsdf_dict(), om.tsdf_dict()
    assert set(g) == set(r) and len(r) > 200

    def digest(d):
        h = hashlib.sha256()
        for key in sorted(d):
            dist, w, rgba, upd = d[key]
            h.update(np.ascontiguousarray(dist).tobytes()); h.update(np.ascontiguousarray(w).tobytes())
            h.update(np.ascontiguousarray(rgba).tobytes()); h.update(bytes([int(upd) & 0xFF]))
        return h.hexdigest()
    if digest(g) != digest(r):
        compare_tsdf(g, r, exact=True)   # names the first differing block
        raise AssertionError("digests differ")


def _mixed_taking_order(n):
    """MixedThreadSafeIndex::getNextIndexImpl (integrator_utils.cc:54-63): point index for every place in the order."""
    s = np.arange(n)
    groups = n // 1024
    if groups == 0:
        return s
    out = (s % groups) * 1024 + s // groups
    out[s >= groups * 1024] = s[s >= groups * 1024]
    return out


@pytest.mark.parametrize("order_mode", [0, 1])
def test_fast_positive_time_budget_takes_a_prefix_of_the_order(oracle, order_mode):
    """max_integration_time_s > 0 (tsdf_integrator.cc:496-499): the threads stop TAKING points when the budget is spent.
    The device decides before the frame how many points the budget pays for (time per point of the earlier calls) and
    takes that prefix of the taking order.  Checked against the oracle fed the same cloud with every point the device
    did not take made invalid (a zero-length ray: isPointValid drops it, tsdf_integrator.cc:84-99) — same order, same
    sets, bit-exact."""
    from voxblox_amd import capi
    voxel = 0.1
    # (sorted order: frames without two points of equal squared norm — std::sort leaves ties unspecified)
    frames = [_small_room(k) for k in range(4)] if order_mode == 0 else [_unique_norm_frame(k) for k in (1, 6, 11)]
    ocfg, gcfg = _cfgs(oracle, 4 * voxel, integration_order_mode=order_mode)
    gcfg.max_integration_time_s = 2e-4          # 200 us: less than a frame costs
    oracle.lib().orc_fast_reset_counter_set(0)
    om = oracle.OracleMap(voxel, 16)
    oi = om.tsdf_integrator("fast", ocfg)
    gm = capi.Map(voxel, 16, max_blocks=4096)
    cut = 0
    for f, (pose, pts, col) in enumerate(frames):
        gm.integrate(capi.TSDF_FAST, gcfg, pose[0], pose[1], pts, col)
        c = gm.counters()
        n, taken = len(pts), c["points_taken"]
        assert c["points"] == n and taken <= n
        if f == 0:
            assert taken == n and c["time_budget_exceeded"] == 1      # nothing measured yet: everything, and it says so
        if order_mode == 0:
            order = _mixed_taking_order(n)
        else:   # SortedThreadSafeIndex: ascending float squaredNorm (Eigen: x*x + (y*y + z*z)), ties by index here
            q = pts.astype(np.float32)
            order = np.argsort(q[:, 0] * q[:, 0] + (q[:, 1] * q[:, 1] + q[:, 2] * q[:, 2]), kind="stable")
        p2 = pts.copy()
        p2[order[taken:]] = 0.0
        cut += n - taken
        oi.integrate(pose[0], pose[1], p2, col)
        compare_tsdf(gm.tsdf_dict(), om.tsdf_dict(), exact=True)
    assert cut > 0                                # the budget did cut frames short


def test_fast_generous_time_budget_changes_nothing(oracle):
    from voxblox_amd import capi
    frames = [_small_room(k) for k in range(3)]
    om, oi, gm = _run(oracle, "fast", 0.1, frames, max_integration_time_s=10.0)
    compare_tsdf(gm.tsdf_dict(), om.tsdf_dict(), exact=True)
    assert gm.counters()["points_taken"] == len(frames[-1][1]) and gm.counters()["time_budget_exceeded"] == 0
