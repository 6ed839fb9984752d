"""-m gpu: host <-> HBM coherence entry points: the bulk block mirror
must return exactly what the per-block call returns, for both layers, and report missing blocks."""
import numpy as np
import pytest

from voxblox_amd import scenes

pytestmark = pytest.mark.gpu


def test_blocks_download_equals_per_block():
    from voxblox_amd import capi
    gm = capi.Map(0.05, 16, max_blocks=2048)
    cfg = capi.tsdf_cfg(default_truncation_distance=0.2)
    for k in (0, 5):
        pose, pts, col = scenes.room_frame(k, 100, f=80.0, width=160, height=120)
        gm.integrate(capi.TSDF_MERGED, cfg, pose[0], pose[1], pts, col)
    ecfg = capi.esdf_cfg(reference_order=0, min_distance_m=0.1)
    gm.esdf_update(ecfg, batch=False, clear_updated_flag=False)
    gm.esdf_add_new_robot_position(capi.esdf_cfg(reference_order=0, min_distance_m=0.1, clear_sphere_radius=0.5, occupied_sphere_radius=1.0),
                                   np.zeros(3, np.float32))
    for layer in (capi.LAYER_TSDF, capi.LAYER_ESDF):
        idx = gm.block_indices(layer)
        assert len(idx) > 50
        v, u, hd = gm.blocks_download(idx, layer)
        assert v.shape == (len(idx), 4096)
        for k in range(0, len(idx), 7):
            v1, u1, hd1 = gm.block_download(idx[k], layer)
            assert v[k].tobytes() == v1.tobytes()
            assert (u[k], hd[k]) == (u1, hd1)
    # page-locked staging buffer
    st = gm.pinned_voxels(len(idx), capi.LAYER_ESDF)
    v2, u2, _ = gm.blocks_download(idx, capi.LAYER_ESDF, out=st)
    assert v2.tobytes() == v.tobytes() and np.array_equal(u2, u)
    # updated-block listing + bulk download is the per-frame mirror of INTEGRATION.md
    upd = gm.blocks_updated(capi.UPDATE_MAP)
    v, u, _ = gm.blocks_download(upd)
    assert len(upd) == gm.num_blocks() and (u & 1).all()
    gm.clear_updated(capi.UPDATE_MAP)
    assert len(gm.blocks_updated(capi.UPDATE_MAP)) == 0
    assert len(gm.blocks_updated(capi.UPDATE_MESH)) == gm.num_blocks()   # other bits untouched
    # a block that is not in the layer is an error, like the per-block call
    with pytest.raises(Exception):
        gm.blocks_download(np.array([[1000, 1000, 1000]], np.int32))
    assert gm.blocks_download(np.zeros((0, 3), np.int32))[0].shape[0] == 0


def test_remove_distant_blocks_matches_reference(oracle):
    """Layer::removeDistantBlocks (layer.h:170-182; caller tsdf_server.cc:315) on both layers:
    same surviving blocks as the oracle, survivors untouched, and integration goes on afterwards."""
    from voxblox_amd import capi
    voxel = 0.1
    gm = capi.Map(voxel, 16, max_blocks=2048)
    om = oracle.OracleMap(voxel, 16)
    gcfg = capi.tsdf_cfg(default_truncation_distance=4 * voxel)
    oi = om.tsdf_integrator("simple", oracle.tsdf_cfg(default_truncation_distance=4 * voxel, integrator_threads=1))
    frames = [scenes.room_frame(k, 100, f=40.0, width=80, height=60) for k in (0, 30, 60)]
    for pose, pts, col in frames[:2]:
        gm.integrate(capi.TSDF_SIMPLE, gcfg, pose[0], pose[1], pts, col)
        oi.integrate(pose[0], pose[1], pts, col)
    ecfg = capi.esdf_cfg(reference_order=0, min_distance_m=2 * voxel)
    gm.esdf_update(ecfg, batch=False, clear_updated_flag=True)
    oe = om.esdf_integrator(oracle.esdf_cfg(min_distance_m=2 * voxel))
    oe.update_from_tsdf_layer(True)
    before = gm.tsdf_dict()
    center = np.array([0.5, -0.2, -1.0], np.float32)
    gm.remove_distant_blocks(center, 2.5, capi.LAYER_TSDF)
    om.remove_distant_blocks(center, 2.5, 0)
    gm.remove_distant_blocks(center, 3.0, capi.LAYER_ESDF)
    om.remove_distant_blocks(center, 3.0, 1)
    got, ref = gm.tsdf_dict(), om.tsdf_dict()
    assert set(got) == set(ref) and 0 < len(got) < len(before)
    for k in got:
        assert np.array_equal(got[k][0], before[k][0]) and np.array_equal(got[k][1], before[k][1])
    ge = {tuple(int(x) for x in i) for i in gm.block_indices(capi.LAYER_ESDF)}
    assert ge == set(om.esdf_dict().keys()) and len(ge) > len(got)     # the layers are independent
    # removed blocks come back as fresh blocks when observed again
    pose, pts, col = frames[2]
    gm.integrate(capi.TSDF_SIMPLE, gcfg, pose[0], pose[1], pts, col)
    oi.integrate(pose[0], pose[1], pts, col)
    got, ref = gm.tsdf_dict(), om.tsdf_dict()
    assert set(got) == set(ref)
    for k in ref:
        assert np.array_equal(got[k][0].view(np.uint32), ref[k][0].view(np.uint32))
        assert np.array_equal(got[k][1].view(np.uint32), ref[k][1].view(np.uint32))


def test_block_upload_round_trip_both_layers():
    """loadMap / tsdfMapCallback path (tsdf_server.cc:566-578, 639-653): blocks written into a
    fresh map through vbx_block_upload read back identically, for both layers, and uploading one
    layer leaves the other layer's block alone."""
    from voxblox_amd import capi
    voxel = 0.1
    src = capi.Map(voxel, 16, max_blocks=1024)
    cfg = capi.tsdf_cfg(default_truncation_distance=4 * voxel)
    pose, pts, col = scenes.room_frame(3, 100, f=40.0, width=80, height=60)
    src.integrate(capi.TSDF_MERGED, cfg, pose[0], pose[1], pts, col)
    src.esdf_update(capi.esdf_cfg(reference_order=0, min_distance_m=2 * voxel), batch=True, clear_updated_flag=False)
    dst = capi.Map(voxel, 16, max_blocks=1024)
    idx = src.block_indices(capi.LAYER_ESDF)
    ev, eu, _ = src.blocks_download(idx, capi.LAYER_ESDF)
    tv, tu, thd = src.blocks_download(idx, capi.LAYER_TSDF)
    for k, i in enumerate(idx):                      # ESDF first: TSDF upload must not disturb it
        dst.block_upload(i, ev[k], int(eu[k]), 0, capi.LAYER_ESDF)
    assert dst.num_blocks(capi.LAYER_TSDF) == 0 and dst.num_blocks(capi.LAYER_ESDF) == len(idx)
    for k, i in enumerate(idx):
        dst.block_upload(i, tv[k], int(tu[k]), int(thd[k]), capi.LAYER_TSDF)
    ev2, eu2, _ = dst.blocks_download(idx, capi.LAYER_ESDF)
    tv2, tu2, thd2 = dst.blocks_download(idx, capi.LAYER_TSDF)
    assert ev2.tobytes() == ev.tobytes() and np.array_equal(eu2, eu)
    assert tv2.tobytes() == tv.tobytes() and np.array_equal(tu2, tu) and np.array_equal(thd2, thd)
    # overwrite of an existing block
    z = np.zeros(4096, capi.ESDF_VOXEL_DTYPE)
    dst.block_upload(idx[0], z, 0, 0, capi.LAYER_ESDF)
    assert dst.block_download(idx[0], capi.LAYER_ESDF)[0].tobytes() == z.tobytes()
    assert dst.block_download(idx[0], capi.LAYER_TSDF)[0].tobytes() == tv[0].tobytes()


def test_stable_radix_sort_selftest():
    """The hand-written stable LSD radix sort (csrc/vbx_sort.hpp) against std::stable_sort: sizes
    around the tile boundaries, every field width class (1..12 bits per pass, 1-6 passes), keys
    only and key/value pairs, random keys and keys with long runs of equal fields."""
    from voxblox_amd import capi
    gm = capi.Map(0.1, 16, max_blocks=64)
    cases = [(0, 32, 52), (1, 32, 52), (63, 0, 64), (64, 44, 64), (2047, 32, 53), (2048, 32, 56), (2049, 32, 57),
             (100_000, 44, 64), (307_200, 32, 53), (1_000_003, 44, 64), (400_000, 32, 56), (50_000, 0, 64),
             (70_000, 7, 8), (70_000, 13, 26), (70_000, 20, 20),
             # tile boundaries of the fused pass (8192 keys per workgroup), its widest field, many tiles
             (8191, 32, 52), (8192, 0, 20), (8193, 32, 52), (16_385, 3, 33), (3_000_001, 32, 57)]
    for n, b, e in cases:
        for seed in (2, 3):
            gm.selftest_sort(n, b, e, seed, with_vals=True)
            gm.selftest_sort(n, b, e, seed + 10, with_vals=False)


def test_stable_radix_sort_beyond_one_lookback_group():
    """More than 512 tiles (4 M keys): the tiles of the fused pass form groups, a group hands its running totals to the
    next (csrc/vbx_sort.hpp) — the Simple integrator's 35-43 M updates and whole-frame replay rounds at fine voxels.
    Sizes around the group boundary, several groups, two and three passes, keys and pairs."""
    from voxblox_amd import capi
    gm = capi.Map(0.1, 16, max_blocks=64)
    for n, b, e in [(512 * 8192, 44, 64), (512 * 8192 + 1, 32, 52), (1024 * 8192 + 77, 32, 58), (9_000_001, 44, 64),
                    (20_000_003, 32, 58)]:
        gm.selftest_sort(n, b, e, 2, with_vals=False)
        gm.selftest_sort(n, b, e, 3, with_vals=(n < 10_000_000))


def test_three_launch_radix_passes(monkeypatch):
    """VBX_SORT_FUSED=0 (read when a handle sorts for the first time) selects count / scan / scatter as three
    launches per pass — the form every sort had before the fused pass, kept for fields wider than 30 bits and
    as the A/B switch."""
    from voxblox_amd import capi
    monkeypatch.setenv("VBX_SORT_FUSED", "0")
    gm = capi.Map(0.1, 16, max_blocks=64)
    for n, b, e in [(1, 32, 52), (2049, 32, 57), (307_200, 32, 53), (1_000_003, 44, 64)]:
        gm.selftest_sort(n, b, e, 4, with_vals=True)
        gm.selftest_sort(n, b, e, 14, with_vals=False)


def test_single_launch_scan_selftest():
    """The library's chained exclusive scan (one launch, decoupled look-back over generation-tagged tile
    words) against a host loop: sizes around the 4096-item tile, the frame-sized cases, millions of
    counters (more tiles than are resident at once), and back-to-back calls reusing the descriptors."""
    from voxblox_amd import capi
    gm = capi.Map(0.1, 16, max_blocks=64)
    for n in (0, 1, 2, 63, 64, 4095, 4096, 4097, 8192, 70001, 307201, 717000, 1 << 20, 5_000_003):
        for seed in (0, 1):
            gm.selftest_scan(n, seed, repeats=3)
    for rep in range(50):     # many short scans in a row: generation tags and the ticket base advance
        gm.selftest_scan(300 + 4096 * (rep % 5), rep, repeats=2)


def test_removed_blocks_free_their_slots_sliding_window(oracle):
    """removeDistantBlocks must free memory like Layer::removeDistantBlocks (layer.h:170-182).  A
    sensor walks through a long corridor of frames while the map keeps only the blocks near it; the pool
    (max_blocks) is far smaller than the number of blocks ever touched, so the stream only survives if
    pool slots and hash entries are recycled — and what is left must equal the oracle doing the same."""
    from voxblox_amd import capi
    from parity_utils import compare_tsdf
    voxel = 0.1
    gm = capi.Map(voxel, 16, max_blocks=160)
    cfg = capi.tsdf_cfg(default_truncation_distance=4 * voxel)
    oracle.lib().orc_fast_reset_counter_set(0)
    om = oracle.OracleMap(voxel, 16)
    oi = om.tsdf_integrator("merged", oracle.tsdf_cfg(default_truncation_distance=4 * voxel, integrator_threads=1))
    touched = set()
    for k in range(40):
        pose, pts, col = scenes.room_frame(k, 100, f=40.0, width=80, height=60)
        pos = pose[0] + np.array([2.5 * k, 0, 0], np.float32)         # the whole scene slides along x
        gm.integrate(capi.TSDF_MERGED, cfg, pos, pose[1], pts, col)
        oi.integrate(pos, pose[1], pts, col)
        touched |= {tuple(int(x) for x in i) for i in gm.block_indices()}
        gm.remove_distant_blocks(pos, 4.0)
        om.remove_distant_blocks(pos, 4.0)
        assert gm.num_blocks() == om.num_blocks(0) <= 160
    assert len(touched) > 3 * 160                                     # far more blocks than the pool ever held at once
    compare_tsdf(gm.tsdf_dict(), om.tsdf_dict(), exact=True)
    gm.clear()                                                        # everything gone: the map is as new
    assert gm.num_blocks() == 0
    pose, pts, col = scenes.room_frame(3, 100, f=40.0, width=80, height=60)
    gm.integrate(capi.TSDF_MERGED, cfg, pose[0], pose[1], pts, col)
    om2 = oracle.OracleMap(voxel, 16)
    om2.tsdf_integrator("merged", oracle.tsdf_cfg(default_truncation_distance=4 * voxel, integrator_threads=1)).integrate(
        pose[0], pose[1], pts, col)
    compare_tsdf(gm.tsdf_dict(), om2.tsdf_dict(), exact=True)


def test_batched_block_upload_round_trip():
    """vbx_blocks_upload (loadMap / tsdfMapCallback path): a whole map copied into a fresh one with one call
    per layer — voxels, updated bits and has_data identical, also for blocks uploaded twice (overwrite) and
    into recycled slots."""
    from voxblox_amd import capi
    pose, pts, col = scenes.room_frame(2, 100, f=40.0, width=80, height=60)
    src = capi.Map(0.1, 16, max_blocks=1024)
    src.integrate(capi.TSDF_FAST, capi.tsdf_cfg(default_truncation_distance=0.4), pose[0], pose[1], pts, col)
    src.esdf_update(capi.esdf_cfg(reference_order=0, min_distance_m=0.2), batch=True, clear_updated_flag=False)
    idx = src.block_indices()
    tv, tu, th = src.blocks_download(idx, capi.LAYER_TSDF)
    eidx = src.block_indices(capi.LAYER_ESDF)
    ev, eu, _ = src.blocks_download(eidx, capi.LAYER_ESDF)
    dst = capi.Map(0.1, 16, max_blocks=1024)
    dst.integrate(capi.TSDF_FAST, capi.tsdf_cfg(default_truncation_distance=0.4), pose[0], pose[1], pts[:500], col[:500])
    dst.clear()                                                       # recycled slots underneath
    for rep in range(2):                                              # the second pass overwrites in place
        dst.blocks_upload(idx, tv, tu, th, capi.LAYER_TSDF)
        dst.blocks_upload(eidx, ev, eu, None, capi.LAYER_ESDF)
    assert dst.num_blocks() == len(idx) and dst.num_blocks(capi.LAYER_ESDF) == len(eidx)
    tv2, tu2, th2 = dst.blocks_download(idx, capi.LAYER_TSDF)
    ev2, eu2, _ = dst.blocks_download(eidx, capi.LAYER_ESDF)
    assert tv2.tobytes() == tv.tobytes() and np.array_equal(tu2, tu) and np.array_equal(th2, th)
    assert ev2.tobytes() == ev.tobytes() and np.array_equal(eu2, eu)


def test_voxel_visit_count_overflow_fails_loudly():
    """the per-ray voxel counts are scanned in 32 bits; a cloud whose rays visit more than 2^32
    voxels in one call must be refused (VBX_ERR_CAPACITY), not wrap and write out of bounds."""
    from voxblox_amd import capi
    gm = capi.Map(0.01, 16, max_blocks=4096)
    cfg = capi.tsdf_cfg(default_truncation_distance=0.04, max_ray_length_m=40.0)
    n = 1_200_000
    pts = np.tile(np.array([[18.0, 17.0, 16.0]], np.float32), (n, 1))       # ~5100 voxels per ray
    col = np.full((n, 4), 255, np.uint8)
    with pytest.raises(capi.VbxError, match="split the cloud"):
        gm.integrate(capi.TSDF_SIMPLE, cfg, np.zeros(3, np.float32), np.array([1, 0, 0, 0], np.float32), pts, col)
    assert gm.num_blocks() == 0
    with pytest.raises(capi.VbxError, match="split the cloud"):              # Fast: the list buffer bound
        big = np.tile(pts, (2, 1)) + np.random.RandomState(0).uniform(-8, 8, (2 * n, 3)).astype(np.float32)
        gm.integrate(capi.TSDF_FAST, cfg, np.zeros(3, np.float32), np.array([1, 0, 0, 0], np.float32),
                     big.astype(np.float32), np.full((2 * n, 4), 255, np.uint8))
