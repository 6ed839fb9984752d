"""-m gpu: the drop-in proof.  voxblox_amd/host/dropin/{tsdf,esdf}_integrator_hip.cc are compiled against
voxblox's OWN headers (include/voxblox/integrator/tsdf_integrator.h:100-103, :219-220, :241-242, :290-291;
esdf_integrator.h:80-107 — unchanged, over the dependency stand-ins of oracle/ref_shims) and linked in
place of src/integrator/tsdf_integrator.cc / esdf_integrator.cc into oracle/_ref/libvbxref_hip.so, next
to the reference's remaining sources (block.cc, integrator_utils.cc, marching_cubes.cc, ...).  The test
harness (oracle/ref_harness.cc) then drives voxblox's real classes — TsdfIntegratorFactory::create,
integratePointCloud on a host Layer<TsdfVoxel>, EsdfIntegrator, and voxblox's own MeshIntegrator<TsdfVoxel> whose
generateMesh is the HIP specialisation of voxblox_amd/host/dropin/mesh_integrator_hip.h (mesh_integrator.h:142-195,
:250-270; round 6 — until round 5 the reference's CPU mesher read the mirrored host layer) — exactly as it drives the
pure-CPU reference build, and the resulting HOST layers and MeshLayers must hash to the golden digests the reference
build produced (tests/golden/reference_digests.json).

The library is built where /root/reference exists (__graft_entry__.build()) and travels to the GPU box
like every other built .so; without it the test fails loudly instead of skipping."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import scenarios as S  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(HERE, "golden", "reference_digests.json")))["scenarios"]

TSDF_BITEXACT = [n for n in sorted(S.SCENARIOS) if S.SCENARIOS[n].get("esdf") is None
                 # "sorted" order: ties between equal squared norms are unspecified in the reference (std::sort)
                 and not S.SCENARIOS[n]["cfg"].get("integration_order_mode")]


@pytest.mark.parametrize("name", TSDF_BITEXACT)
def test_real_voxblox_classes_over_hip_reproduce_reference_digest(oracle, name):
    """TSDF integrators (and, in the mesh_* scenarios, voxblox's own MeshIntegrator<TsdfVoxel> class with the device
    mesher underneath — generateMesh(true, true) after every frame or one full pass, with and without colours — filling the
    caller's MeshLayer): host Layer after every scenario == the reference's, bit for bit — distances, weights, colours,
    updated bits (kMesh cleared by the mesher included) — and every Mesh of the MeshLayer == the reference's: vertices,
    normals, colours, indices, the `updated` flag, the MeshLayer's own block order."""
    L = oracle.ref_hip_lib()
    assert L.orc_dropin_mesher() == 1, "libvbxref_hip.so was built without the mesher's specialisation"
    m = S.run_on_oracle_api(oracle, L, S.SCENARIOS[name])
    assert S.digest_tsdf(m.tsdf_dict()) == GOLD[name]["tsdf"]
    if "mesh" in GOLD[name]:
        assert S.digest_mesh(m.mesh.as_dict()) == GOLD[name]["mesh"]
        assert oracle.timing_get(L, "mesh/generate")[0] > 0      # the timer of generateMeshOnDevice: the HIP path ran


def test_mesher_dropin_follows_host_side_kmesh_bits(oracle):
    """The block list of generateMesh(only_mesh_updated_blocks = true) is the HOST layer's (mesh_integrator.h:147-152).
    Somebody clears kMesh on the host between two calls (another consumer meshed those blocks): the device still carries
    the bits, the drop-in must not re-mesh those blocks — their Mesh::updated stays false like in the CPU build."""
    H, R = oracle.ref_hip_lib(), oracle.ref_lib()
    sc = S.SCENARIOS["mesh_fast_0p05_incremental"]
    frames = S.frames(sc["n"])
    import ctypes as C
    out = []
    for L in (H, R):
        L.orc_fast_reset_counter_set(0)
        m = oracle.OracleMap(sc["voxel"], 16, L=L)
        c = oracle.TsdfCfg()
        L.orc_tsdf_cfg_default(C.byref(c))
        c.default_truncation_distance = 4 * sc["voxel"]
        c.integrator_threads = 1
        it = m.tsdf_integrator("fast", c)
        ml = m.mesh_layer()
        for k, (pose, pts, col) in enumerate(frames):
            it.integrate(pose[0], pose[1], pts, col)
            if k == 1:   # the host clears kMesh (bit 1) on half of the updated blocks, voxels untouched
                d = m.tsdf_dict()
                for j, key in enumerate(sorted(d)):
                    if j % 2 == 0 and (d[key][3] & 2):
                        m.tsdf_block_set(key, d[key][0], d[key][1], d[key][2], d[key][3] & ~2)
            ml.clear_updated()
            ml.generate(only_mesh_updated_blocks=True, clear_updated_flag=True, use_color=True, min_weight=1e-4)
        out.append((S.digest_tsdf(m.tsdf_dict()), S.digest_mesh(ml.as_dict()), {k: v["updated"] for k, v in ml.as_dict().items()}))
    assert out[0][2] == out[1][2], "the set of re-meshed blocks differs from the CPU build's"
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1]


def test_real_voxblox_esdf_class_over_hip(oracle):
    """EsdfIntegrator through the real class, drop-in default (the reference's own queue order).  The reference's result
    depends on the order in which Layer::getAllAllocatedBlocks / getAllUpdatedBlocks list the blocks — the iteration order of
    the host Layer's std::unordered_map, i.e. the sequence in which blocks were inserted.  The drop-in inserts a frame's new
    blocks in the sequence the reference's single-threaded integrator does (temp_block_map_'s iteration order,
    tsdf_integrator.cc:107-147, replayed from the first touch of every new block: vbx_blocks_new_ordered), so both Layers
    iterate alike and both scenarios must come out bit for bit: batch (min_diff_m 0, UNSWITCHED sign-mismatch rule) and
    incremental (default Config) — distances, flags, parents, updated bits, golden digests."""
    L, R = oracle.ref_hip_lib(), oracle.ref_lib()
    assert L.vbx_dropin_get_esdf_reference_order() == 1
    name = "esdf_batch_min_diff0"
    g = S.run_on_oracle_api(oracle, L, S.SCENARIOS[name]).esdf_dict()
    r = S.run_on_oracle_api(oracle, R, S.SCENARIOS[name]).esdf_dict()
    assert S.digest_esdf(r) == GOLD[name]["esdf"]
    assert set(g) == set(r)
    n_par = n = 0
    for k in r:
        assert np.array_equal(g[k][0].view(np.uint32), r[k][0].view(np.uint32)), k
        assert np.array_equal(g[k][1], r[k][1]) and g[k][3] == r[k][3], k
        n_par += int((np.asarray(g[k][2]).reshape(-1, 3) != np.asarray(r[k][2]).reshape(-1, 3)).any(axis=1).sum())
        n += int((r[k][1] & 1).sum())
    assert n > 10000 and n_par == 0, (n_par, n)
    # incremental, default Config: since round 5 the drop-in allocates the blocks an integrate call added in the sequence the
    # reference's single-threaded integrator inserts them (vbx_blocks_new_ordered), so the host Layer iterates like the CPU
    # run's, the walk order of every update is the reference's, and the result is the reference's — digest included
    name = "esdf_incremental"
    gm = S.run_on_oracle_api(oracle, L, S.SCENARIOS[name])
    rm = S.run_on_oracle_api(oracle, R, S.SCENARIOS[name])
    assert [tuple(b) for b in gm.block_indices(0)] == [tuple(b) for b in rm.block_indices(0)]   # Layer<TsdfVoxel> iteration order
    g, r = gm.esdf_dict(), rm.esdf_dict()
    assert S.digest_esdf(r) == GOLD[name]["esdf"]
    assert set(g) == set(r)
    n = nd = 0
    for k in r:
        assert np.array_equal(g[k][1], r[k][1]) and g[k][3] == r[k][3], k
        obs = (r[k][1] & 1).astype(bool)
        n += int(obs.sum())
        nd += int((g[k][0].view(np.uint32)[obs] != r[k][0].view(np.uint32)[obs]).sum())
    assert n > 10000 and nd == 0, (nd, n)
    assert S.digest_esdf(g) == GOLD[name]["esdf"]


def test_real_voxblox_esdf_class_over_hip_order_free(oracle):
    """vbx_dropin_set_esdf_reference_order(0): flags and updated bits of the host Layer<EsdfVoxel> identical to the
    reference's golden batch scenario, distances bit-exact against the order-free form of the sign-mismatch rule (see
    test_hip_esdf_batch_against_reference_golden); and the incremental default-config scenario inside the reference's
    own envelope (test_sdf_integrators.cc:270)."""
    L = oracle.ref_hip_lib()
    L.vbx_dropin_set_esdf_reference_order(0)
    try:
        _order_free_checks(oracle, L)
    finally:
        L.vbx_dropin_set_esdf_reference_order(1)


def _order_free_checks(oracle, L):
    name = "esdf_batch_min_diff0"
    sc = S.SCENARIOS[name]
    g = S.run_on_oracle_api(oracle, L, sc).esdf_dict()
    sw = dict(sc, esdf=dict(sc["esdf"], cfg=dict(sc["esdf"]["cfg"], oracle_orderfree_sign_mismatch=1)))
    o = S.run_on_oracle_api(oracle, oracle.lib(), sw).esdf_dict()
    assert set(g) == set(o) and len(g) == GOLD[name]["esdf"]["blocks"]
    for k in o:
        assert np.array_equal(g[k][1], o[k][1]) and g[k][3] == o[k][3], k
        assert np.array_equal(g[k][0].view(np.uint32), o[k][0].view(np.uint32)), k
    name = "esdf_incremental"
    sc = S.SCENARIOS[name]
    g = S.run_on_oracle_api(oracle, L, sc).esdf_dict()
    r = S.run_on_oracle_api(oracle, oracle.lib(), sc).esdf_dict()
    assert set(g) == set(r)
    se = n = n_fix_diff = 0
    for k in r:
        assert np.array_equal(g[k][1] & 1, r[k][1] & 1), k           # observed mask
        # `fixed` is only rewritten when the TSDF value moved by more than min_diff_m against the stored
        # ESDF value (esdf_integrator.cc:221-255); the device wavefront stores the exact fixed point where
        # the reference stores a value up to min_diff_m short of it, so a TSDF change inside that slack can
        # flip the flag on one side only — a voxel or two per map, each within min_diff_m of the reference
        fx = ((g[k][1] ^ r[k][1]) & 8) != 0
        if fx.any():
            n_fix_diff += int(fx.sum())
            assert np.abs(g[k][0][fx] - r[k][0][fx]).max() <= 1e-3, k
        obs = (r[k][1] & 1).astype(bool)
        se += float(((g[k][0][obs] - r[k][0][obs]) ** 2).sum())
        n += int(obs.sum())
    assert n > 10000 and (se / n) ** 0.5 < 1e-2 and n_fix_diff <= 1e-4 * n


def test_layer_reuse_and_clear_through_the_real_classes(oracle):
    """Host-side lifetime events the association table must survive: two maps alive at once, a map
    destroyed and another created (possibly at the same address), removeAllBlocks() between frames."""
    L = oracle.ref_hip_lib()
    sc = S.SCENARIOS["fast_default"]
    a = S.run_on_oracle_api(oracle, L, sc)
    b = S.run_on_oracle_api(oracle, L, S.SCENARIOS["merged_default"])
    assert S.digest_tsdf(a.tsdf_dict()) == GOLD["fast_default"]["tsdf"]
    assert S.digest_tsdf(b.tsdf_dict()) == GOLD["merged_default"]["tsdf"]
    del a, b
    for _ in range(3):
        m = S.run_on_oracle_api(oracle, L, sc)
        assert S.digest_tsdf(m.tsdf_dict()) == GOLD["fast_default"]["tsdf"]
        m.clear(0)                                                   # Layer::removeAllBlocks on the host
        assert m.num_blocks(0) == 0
        del m
    # removeAllBlocks() between two frames: the second frame must land in an empty map on the device too
    import ctypes as C
    frames = S.frames(2)
    def cfg_of():
        c = oracle.TsdfCfg()
        L.orc_tsdf_cfg_default(C.byref(c))
        c.default_truncation_distance = 4 * sc["voxel"]
        c.integrator_threads = 1
        return c
    m1 = oracle.OracleMap(sc["voxel"], 16, L=L)
    i1 = m1.tsdf_integrator("fast", cfg_of())
    i1.integrate(frames[0][0][0], frames[0][0][1], frames[0][1], frames[0][2])
    m1.clear(0)
    i1.integrate(frames[1][0][0], frames[1][0][1], frames[1][1], frames[1][2])
    m2 = oracle.OracleMap(sc["voxel"], 16, L=L)
    i2 = m2.tsdf_integrator("fast", cfg_of())
    i2.integrate(frames[1][0][0], frames[1][0][1], frames[1][1], frames[1][2])
    assert S.digest_tsdf(m1.tsdf_dict()) == S.digest_tsdf(m2.tsdf_dict())
