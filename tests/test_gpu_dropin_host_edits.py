"""-m gpu: host-side Layer edits between calls reach the device map of the drop-in without any change to the
caller ("callers read/mutate the same host Layer directly between calls").  Everything here goes
through voxblox's REAL classes and containers (oracle/_ref/libvbxref_hip.so = the reference's headers and
remaining sources + voxblox_amd/host/dropin/*.cc) and is compared with the pure-CPU build of the same code
doing the same edits:

  * Layer::removeDistantBlocks after every frame            (layer.h:170-182; caller tsdf_server.cc:315)
  * a TSDF layer that was LOADED into the host Layer, then EsdfIntegrator::updateFromTsdfLayerBatch and
    further integration                                      (io::LoadBlocksFromFile, tsdf_server.cc:566-578;
                                                              esdf_server / tsdf_to_esdf)
  * an in-place overwrite of an existing block without any marker (deserializeMsgToLayer kUpdate,
    conversions_inl.h:80-88), down to a single voxel: caught by the full fingerprint of the voxel array
  * Update bits set by the host on an existing block (Block::mergeBlock, block_inl.h:120)
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import scenarios as S  # noqa: E402

pytestmark = pytest.mark.gpu
VOXEL = 0.1


@pytest.fixture(autouse=True, params=[-1, 0], ids=["reconcile_touched", "reconcile_every_line"])
def reconcile_mode(request, oracle):
    """Every test of this file runs in both reconcile modes of the drop-in (device_mirror.h): -1 = O(touched), the default
    since round 6 (the mirror's own records + a rotating window; an in-place overwrite WITHOUT any marker has to be announced
    with hip::markLayerEdited), 0 = every line of every host block on every call (round 5's default: sees a single-voxel
    poke without being told)."""
    H = oracle.ref_hip_lib()
    H.vbx_dropin_set_reconcile_mode(request.param)
    yield request.param
    H.vbx_dropin_set_reconcile_mode(-1)


def _cfg(oracle, L):
    c = oracle.TsdfCfg()
    L.orc_tsdf_cfg_default(C.byref(c))
    c.default_truncation_distance = 4 * VOXEL
    c.integrator_threads = 1
    return c


def _cpu_lib(oracle):
    """The checker: the reference's own sources where they are built (oracle/_ref/libvbxref.so), else the restatement."""
    return oracle.ref_lib() if oracle.ref_available() else oracle.lib()


def _same_tsdf(a, b):
    da, db = a.tsdf_dict(), b.tsdf_dict()
    assert set(da) == set(db), (len(da), len(db), sorted(set(da) ^ set(db))[:4])
    for k in da:
        assert np.array_equal(da[k][0].view(np.uint32), db[k][0].view(np.uint32)), k
        assert np.array_equal(da[k][1].view(np.uint32), db[k][1].view(np.uint32)), k
        assert np.array_equal(da[k][2], db[k][2]), k
    return len(da)


@pytest.mark.parametrize("kind", ["fast", "merged"])
def test_remove_distant_blocks_after_every_frame(oracle, kind):
    """tsdf_server's loop: integratePointcloud, then removeDistantBlocks around the sensor.  The device must drop
    the same blocks (a block that comes back later starts from weight 0, and its pool slot is free meanwhile)."""
    H, R = oracle.ref_hip_lib(), _cpu_lib(oracle)
    maps = []
    for L in (H, R):
        L.orc_fast_reset_counter_set(0)
        m = oracle.OracleMap(VOXEL, 16, L=L)
        it = m.tsdf_integrator(kind, _cfg(oracle, L))
        for pose, pts, col in S.frames(6, step=9):
            it.integrate(pose[0], pose[1], pts, col)
            m.remove_distant_blocks(pose[0], 2.6)
        maps.append((m, it))
    n = _same_tsdf(maps[0][0], maps[1][0])
    st = maps[0][0].dropin_stats()
    assert n > 10 and st["removed_blocks"] > 10, (n, st)      # blocks really left, on the device too
    assert st["uploaded_blocks"] == 0, st                      # and nothing was re-uploaded: the mirror's own writes are not "edits"


def _loaded_pair(oracle, n_frames=3):
    """(hip map filled through the HOST Layer only, cpu map) holding the same TSDF layer."""
    H, R = oracle.ref_hip_lib(), _cpu_lib(oracle)
    R.orc_fast_reset_counter_set(0)
    src = oracle.OracleMap(VOXEL, 16, L=R)
    it = src.tsdf_integrator("simple", _cfg(oracle, R))
    for pose, pts, col in S.frames(n_frames):
        it.integrate(pose[0], pose[1], pts, col)
    dst = oracle.OracleMap(VOXEL, 16, L=H)
    for k, (d, w, c, _) in src.tsdf_dict().items():
        dst.tsdf_block_set(k, d, w, c, 7)      # Layer::addBlockFromProto sets all Update bits (layer_inl.h:227)
        src.tsdf_block_set(k, d, w, c, 7)
    return dst, src


def _same_esdf_bits(g, o, min_blocks=20):
    assert set(g) == set(o) and len(g) > min_blocks
    for k in o:
        assert np.array_equal(g[k][1], o[k][1]) and g[k][3] == o[k][3], k                      # flags, updated bits
        assert np.array_equal(g[k][0].view(np.uint32), o[k][0].view(np.uint32)), k            # distances, bit for bit
        assert np.array_equal(g[k][2], o[k][2]), k                                             # parents


def test_esdf_batch_over_a_loaded_tsdf_layer(oracle):
    """esdf_server / tsdf_to_esdf: the TSDF layer comes from a file, no integrator ever ran on it.  The ESDF drop-in
    must see it (it used to see an empty device map).  Drop-in default (the reference's own order, blocks visited in the
    iteration order of the host Layer's container): the host Layer<EsdfVoxel> is IDENTICAL to what the CPU build leaves
    in a Layer filled the same way — distances bit for bit, flags, parents, updated bits."""
    dst, src = _loaded_pair(oracle)
    n_blocks = dst.num_blocks(0)

    def esdf_cfg(L):
        c = oracle.EsdfCfg()
        L.orc_esdf_cfg_default(C.byref(c))
        c.min_distance_m = 2 * VOXEL
        c.min_diff_m = 0.0
        return c

    H, R = oracle.ref_hip_lib(), _cpu_lib(oracle)
    assert H.vbx_dropin_get_esdf_reference_order() == 1
    dst.esdf_integrator(esdf_cfg(H)).update_from_tsdf_layer_batch()
    st = dst.dropin_stats()
    assert st["uploaded_blocks"] == n_blocks > 20, (st, n_blocks)
    # the reference's result depends on the order in which getAllAllocatedBlocks lists the blocks, i.e. on how the
    # Layer's unordered_map was filled: the CPU-build Layer is filled exactly like dst (same keys, same sequence)
    ref = oracle.OracleMap(VOXEL, 16, L=R)
    for k, (d, w, c, _) in src.tsdf_dict().items():
        ref.tsdf_block_set(k, d, w, c, 7)
    ref.esdf_integrator(esdf_cfg(R)).update_from_tsdf_layer_batch()
    _same_esdf_bits(dst.esdf_dict(), ref.esdf_dict())


def test_esdf_batch_order_free_switch(oracle):
    """vbx_dropin_set_esdf_reference_order(0): the order-free fixed point instead — flags and updated bits equal to the CPU
    build's, distances bit-exact against the order-free form of the sign-mismatch rule (as in
    test_real_voxblox_esdf_class_over_hip)."""
    H = oracle.ref_hip_lib()
    H.vbx_dropin_set_esdf_reference_order(0)
    try:
        dst, src = _loaded_pair(oracle)

        def esdf_cfg(L):
            c = oracle.EsdfCfg()
            L.orc_esdf_cfg_default(C.byref(c))
            c.min_distance_m = 2 * VOXEL
            c.min_diff_m = 0.0
            return c

        dst.esdf_integrator(esdf_cfg(H)).update_from_tsdf_layer_batch()
        # checker: the restatement with the order-free sign-mismatch switch, on the same TSDF
        chk = oracle.OracleMap(VOXEL, 16)
        for k, (d, w, c, _) in src.tsdf_dict().items():
            chk.tsdf_block_set(k, d, w, c, 7)
        oc = esdf_cfg(oracle.lib())
        oc.oracle_orderfree_sign_mismatch = 1
        chk.esdf_integrator(oc).update_from_tsdf_layer_batch()
        g, o = dst.esdf_dict(), chk.esdf_dict()
        assert set(g) == set(o) and len(g) > 20
        for k in o:
            assert np.array_equal(g[k][1], o[k][1]) and g[k][3] == o[k][3], k
            assert np.array_equal(g[k][0].view(np.uint32), o[k][0].view(np.uint32)), k
    finally:
        H.vbx_dropin_set_esdf_reference_order(1)


def test_integration_continues_on_a_loaded_layer(oracle):
    """loadMap, then the sensor keeps running: the frames must fold into the LOADED voxels (weights, clamps), not
    into an empty device map that then overwrites the host's blocks."""
    dst, src = _loaded_pair(oracle)
    H, R = oracle.ref_hip_lib(), _cpu_lib(oracle)
    frames = S.frames(6)[3:]
    for L, m in ((H, dst), (R, src)):
        L.orc_fast_reset_counter_set(0)
        it = m.tsdf_integrator("fast", _cfg(oracle, L))
        for pose, pts, col in frames:
            it.integrate(pose[0], pose[1], pts, col)
    assert _same_tsdf(dst, src) > 20


def test_in_place_overwrite_of_an_existing_block_is_seen(oracle, reconcile_mode):
    """deserializeMsgToLayer(kUpdate) writes new voxels into a block the layer already has and marks nothing
    (conversions_inl.h:80-88).  The sampled fingerprint of the voxel array tells; the next frame then folds into
    the overwritten values on both sides."""
    H, R = oracle.ref_hip_lib(), _cpu_lib(oracle)
    frames = S.frames(4)
    maps = []
    for L in (H, R):
        L.orc_fast_reset_counter_set(0)
        m = oracle.OracleMap(VOXEL, 16, L=L)
        it = m.tsdf_integrator("merged", _cfg(oracle, L))
        for pose, pts, col in frames[:2]:
            it.integrate(pose[0], pose[1], pts, col)
        d = m.tsdf_dict()
        victims = sorted(k for k in d if (d[k][1] > 0).sum() > 3000)[:3]   # well-filled blocks: every sampled line moves
        assert victims
        for k in victims:   # a different map's values for the same block: halve the distances, double the weights
            dist, w, c, bits = d[k]
            m.tsdf_block_set(k, dist * np.float32(0.5), w * np.float32(2.0), c[:, ::-1].copy(), bits)   # bits untouched
        if reconcile_mode < 0:
            m.dropin_mark_edited(0)   # hip::markLayerEdited: what a caller of deserializeMsgToLayer(kUpdate) adds in this mode
        for pose, pts, col in frames[2:]:
            it.integrate(pose[0], pose[1], pts, col)
        maps.append(m)
    assert _same_tsdf(maps[0], maps[1]) > 10
    st = maps[0].dropin_stats()
    assert st["uploaded_blocks"] == 3, st


def test_a_single_voxel_poke_without_any_marker_is_seen(oracle, reconcile_mode):
    """The round-3/4 hole: the fingerprint sampled 8 of a block's 768 lines, so a write into ONE voxel between two sampled
    lines, with no Update bit, was overwritten by the next mirror.  The default fingerprint now covers every line: a poke
    into a single voxel (here: one whose 12 bytes lie in a line the old sampling never read) is uploaded before the next
    frame integrates, and both sides fold the next frames into the poked value."""
    H, R = oracle.ref_hip_lib(), _cpu_lib(oracle)
    frames = S.frames(4)
    maps = []
    for L in (H, R):
        L.orc_fast_reset_counter_set(0)
        m = oracle.OracleMap(VOXEL, 16, L=L)
        it = m.tsdf_integrator("merged", _cfg(oracle, L))
        for pose, pts, col in frames[:2]:
            it.integrate(pose[0], pose[1], pts, col)
        d = m.tsdf_dict()
        victim = sorted(k for k in d if (d[k][1] > 0).sum() > 3000)[0]
        dist, w, c, bits = d[victim]
        sampled = {(k * 768) // 8 + 48 for k in range(8)}                      # the lines rounds 3-4 looked at
        v = next(i for i in range(100, 4096) if w[i] > 0 and (12 * i) // 64 not in sampled and (12 * i + 11) // 64 not in sampled)
        dist = dist.copy(); w = w.copy()
        dist[v] = np.float32(-0.123); w[v] = np.float32(7.5)
        m.tsdf_block_set(victim, dist, w, c, bits)                              # bits untouched
        if reconcile_mode < 0:
            m.dropin_mark_edited(0)   # O(touched) mode: the poke is announced (no marker at all: mode 0, the other run of this test)
        for pose, pts, col in frames[2:]:
            it.integrate(pose[0], pose[1], pts, col)
        maps.append(m)
    assert _same_tsdf(maps[0], maps[1]) > 10
    st = maps[0].dropin_stats()
    assert st["uploaded_blocks"] == 1, st


def test_update_bits_set_by_the_host_trigger_an_upload(oracle):
    """Block::mergeBlock / addBlockFromProto set updated() (block_inl.h:120, layer_inl.h:227).  After a consumer has
    cleared a bit (the mesher clears kMesh), a bit that is back means the host wrote the block."""
    H = oracle.ref_hip_lib()
    H.orc_fast_reset_counter_set(0)
    m = oracle.OracleMap(VOXEL, 16, L=H)
    it = m.tsdf_integrator("fast", _cfg(oracle, H))
    frames = S.frames(3)
    it.integrate(frames[0][0][0], frames[0][0][1], frames[0][1], frames[0][2])
    ml = m.mesh_layer()
    ml.generate(True, True)                      # the reference's CPU mesher clears kMesh on the host blocks
    it.integrate(frames[1][0][0], frames[1][0][1], frames[1][1], frames[1][2])
    assert m.dropin_stats()["uploaded_blocks"] == 0
    ml.generate(True, True)
    d = m.tsdf_dict()
    k = sorted(d)[0]
    assert (d[k][3] & 2) == 0                    # kMesh is clear on the host
    empty = (np.zeros((0, 3), np.float32), np.zeros((0, 4), np.uint8))
    it.integrate(frames[2][0][0], frames[2][0][1], *empty)   # any drop-in call: the mirror learns which bits the consumers cleared
    assert m.dropin_stats()["uploaded_blocks"] == 0
    m.tsdf_block_set(k, d[k][0], d[k][1], d[k][2], 7)   # same voxels, all bits set again
    it.integrate(frames[2][0][0], frames[2][0][1], frames[2][1], frames[2][2])
    assert m.dropin_stats()["uploaded_blocks"] == 1


def test_loaded_esdf_layer_is_uploaded_and_updated_incrementally(oracle):
    """esdf_server::loadMap loads BOTH layers into the host maps and then keeps updating incrementally.  The ESDF
    drop-in must take the host's ESDF blocks (reconcileEsdfFromHost -> vbx_blocks_upload of the 20-byte EsdfVoxel AoS)
    as the state the next incremental update starts from: afterwards the observed masks equal the CPU build's and the
    distances lie inside the reference's own incremental envelope (the default device wavefront is order-free,
    test_sdf_integrators.cc:270), exactly as for a layer the drop-in computed itself."""
    H, R = oracle.ref_hip_lib(), _cpu_lib(oracle)
    frames = S.frames(6)

    def esdf_cfg(L):
        c = oracle.EsdfCfg()
        L.orc_esdf_cfg_default(C.byref(c))
        c.min_distance_m = 2 * VOXEL
        return c

    # source of the "file": three frames through the CPU build, TSDF + incremental ESDF
    R.orc_fast_reset_counter_set(0)
    src = oracle.OracleMap(VOXEL, 16, L=R)
    it = src.tsdf_integrator("merged", _cfg(oracle, R))
    es = src.esdf_integrator(esdf_cfg(R))
    for pose, pts, col in frames[:3]:
        it.integrate(pose[0], pose[1], pts, col)
        es.update_from_tsdf_layer(True)
    t_src, e_src = src.tsdf_dict(), src.esdf_dict()
    maps = []
    for L in (H, R):   # two fresh maps filled in the same order, then the same three frames on each
        L.orc_fast_reset_counter_set(0)
        m = oracle.OracleMap(VOXEL, 16, L=L)
        for k in sorted(t_src):
            m.tsdf_block_set(k, t_src[k][0], t_src[k][1], t_src[k][2], 3)          # kEsdf clear: nothing pending
        for k in sorted(e_src):
            m.esdf_block_set(k, e_src[k][0], e_src[k][1], e_src[k][2], e_src[k][3])
        ti = m.tsdf_integrator("merged", _cfg(oracle, L))
        ei = m.esdf_integrator(esdf_cfg(L))
        for pose, pts, col in frames[3:]:
            ti.integrate(pose[0], pose[1], pts, col)
            ei.update_from_tsdf_layer(True)
        maps.append(m)
    assert maps[0].dropin_stats()["uploaded_blocks"] >= len(t_src) + len(e_src)
    _same_tsdf(maps[0], maps[1])
    g, r = maps[0].esdf_dict(), maps[1].esdf_dict()
    assert set(g) == set(r) and len(g) >= len(e_src)
    se = n = 0
    for k in r:
        assert np.array_equal(g[k][1] & 1, r[k][1] & 1), k
        obs = (r[k][1] & 1).astype(bool)
        se += float(((g[k][0][obs] - r[k][0][obs]) ** 2).sum())
        n += int(obs.sum())
    # (not the same bits: the frames insert new blocks into the two host Layers in different sequences — the mirror's vs the
    # CPU integrator's — so getAllUpdatedBlocks walks them in different orders, and the reference's result depends on that walk)
    assert n > 10000 and (se / n) ** 0.5 < 1e-2, (n, (se / max(n, 1)) ** 0.5)


def test_more_live_layers_than_the_mirror_table_holds(oracle):
    """The association table keeps 8 device maps (device_mirror.h).  Eleven layers are alive at once here; the first
    one loses its device map on the way and gets it back from the host Layer (which is coherent after every call)
    when it is used again: the result is the CPU build's, and the re-upload shows in the mirror's counters."""
    H, R = oracle.ref_hip_lib(), _cpu_lib(oracle)
    frames = S.frames(2)
    maps = []
    for i in range(11):
        m = oracle.OracleMap(VOXEL, 16, L=H)
        it = m.tsdf_integrator("simple", _cfg(oracle, H))
        it.integrate(frames[0][0][0], frames[0][0][1], frames[0][1], frames[0][2])
        maps.append((m, it))
    assert maps[0][0].dropin_stats()["uploaded_blocks"] == 0
    m0, it0 = maps[0]
    it0.integrate(frames[1][0][0], frames[1][0][1], frames[1][1], frames[1][2])
    ref = oracle.OracleMap(VOXEL, 16, L=R)
    rit = ref.tsdf_integrator("simple", _cfg(oracle, R))
    for pose, pts, col in frames:
        rit.integrate(pose[0], pose[1], pts, col)
    n = _same_tsdf(m0, ref)
    assert n > 10
    assert m0.dropin_stats()["uploaded_blocks"] > 10, m0.dropin_stats()   # the first frame came back from the host
    # the last map never left the table
    m10, it10 = maps[10]
    it10.integrate(frames[1][0][0], frames[1][0][1], frames[1][1], frames[1][2])
    assert _same_tsdf(m10, ref) == n
    assert m10.dropin_stats()["uploaded_blocks"] == 0


def test_robot_position_spheres_in_reference_order_through_voxblox_classes(oracle):
    """EsdfIntegrator::addNewRobotPosition + updateFromTsdfLayer through voxblox's real classes (esdf_server.cc:219-230)
    with the drop-in's default, the reference's order: the sphere pushes wait in raise_ / open_ in the iteration order
    of the reference's HierarchicalIndexMap, updated_blocks_ is the class's own IndexSet.  Two Layers filled the same
    way give the same bits, twice in a row (the second position meets hallucinated voxels), and clear() forgets the
    queued work on the device like it does on the host."""
    H, R = oracle.ref_hip_lib(), _cpu_lib(oracle)
    assert H.vbx_dropin_get_esdf_reference_order() == 1
    dst, src = _loaded_pair(oracle)
    ref = oracle.OracleMap(VOXEL, 16, L=R)
    for k, (d, w, c, _) in src.tsdf_dict().items():
        ref.tsdf_block_set(k, d, w, c, 7)

    def esdf_cfg(L):
        c = oracle.EsdfCfg()
        L.orc_esdf_cfg_default(C.byref(c))
        c.min_distance_m = 2 * VOXEL
        c.clear_sphere_radius = 0.6
        c.occupied_sphere_radius = 1.5
        return c

    ge, re_ = dst.esdf_integrator(esdf_cfg(H)), ref.esdf_integrator(esdf_cfg(R))
    p0 = np.float32(S.frames(1)[0][0][0])
    for step, p in enumerate((p0, p0 + np.float32([0.3, 0.1, 0.0]))):
        for e in (ge, re_):
            e.add_new_robot_position(p)
        _same_esdf_bits(dst.esdf_dict(), ref.esdf_dict(), min_blocks=8)
        for e in (ge, re_):
            e.update_from_tsdf_layer(True)
        _same_esdf_bits(dst.esdf_dict(), ref.esdf_dict())
    # clear() between the spheres and the update: the voxel changes stay, the queued work is gone
    for e in (ge, re_):
        e.add_new_robot_position(p0 + np.float32([-0.4, 0.2, 0.1]))
        e.clear()
        e.update_from_tsdf_layer(True)
    _same_esdf_bits(dst.esdf_dict(), ref.esdf_dict())
