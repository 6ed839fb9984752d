"""CPU: the C-ABI library loads and exports every symbol include/vbx_hip.h declares; without
a GPU the product path fails loudly instead of falling back to anything."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "vbx_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(vbx_[a-z_0-9]+)\s*\(", hdr)))


def test_header_symbols_all_exported():
    from voxblox_amd import capi
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = C.CDLL(capi.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/vbx_hip.h but not exported"
    assert set(capi.EXPORTED_SYMBOLS) == set(declared)


def test_config_defaults_match_reference():
    """tsdf_integrator.h:59-86 and esdf_integrator.h:37-77 defaults."""
    from voxblox_amd import capi
    t = capi.tsdf_cfg()
    assert abs(t.default_truncation_distance - 0.1) < 1e-7 and t.max_weight == 10000.0
    assert t.voxel_carving_enabled == 1 and abs(t.min_ray_length_m - 0.1) < 1e-7
    assert t.max_ray_length_m == 5.0 and t.use_const_weight == 0 and t.allow_clear == 1
    assert t.use_weight_dropoff == 1 and t.use_sparsity_compensation_factor == 0
    assert t.sparsity_compensation_factor == 1.0 and t.integration_order_mode == 0
    assert t.enable_anti_grazing == 0 and t.start_voxel_subsampling_factor == 2.0
    assert t.max_consecutive_ray_collisions == 2 and t.clear_checks_every_n_frames == 1
    e = capi.esdf_cfg()
    assert e.full_euclidean_distance == 0 and e.max_distance_m == 2.0
    assert abs(e.min_distance_m - 0.2) < 1e-7 and e.default_distance_m == 2.0
    assert abs(e.min_diff_m - 0.001) < 1e-9 and e.num_buckets == 20 and e.multi_queue == 0
    assert e.add_occupied_crust == 0 and e.clear_sphere_radius == 1.5 and e.occupied_sphere_radius == 5.0


def test_oracle_and_capi_defaults_agree(oracle):
    from voxblox_amd import capi
    o, g = oracle.tsdf_cfg(), capi.tsdf_cfg()
    for name, _ in capi.TsdfCfg._fields_:
        if name in ("integrator_threads", "merged_bundle_order", "fast_observed_set"):
            continue   # the last two are HIP-only fields; 0 = the reference's semantics
        assert getattr(o, name) == getattr(g, name), name
    assert g.merged_bundle_order == 0 and g.fast_observed_set == 0
    oe, ge = oracle.esdf_cfg(), capi.esdf_cfg()
    for name, _ in capi.EsdfCfg._fields_:
        if name == "reference_order":
            continue   # HIP-only field; 1 (default) = the reference's own queue order, 0 = the order-free fast mode
        assert getattr(oe, name) == getattr(ge, name), name
    assert ge.reference_order == 1
    assert oe.oracle_orderfree_sign_mismatch == 0  # oracle-only switch defaults to the reference


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from voxblox_amd import capi
    with pytest.raises(capi.VbxError) as ei:
        capi.Map(0.05, 16)
    assert "no CPU fallback" in str(ei.value) or "HIP" in str(ei.value)


def test_bad_geometry_rejected():
    from voxblox_amd import capi
    for vs, vps in ((0.0, 16), (-1.0, 16), (0.05, 12), (0.05, 64), (0.05, 2)):
        with pytest.raises(capi.VbxError):
            capi.Map(vs, vps)


def test_factory_error_behaviour():
    """TsdfIntegratorFactory::create: unknown / empty names and a null layer are fatal
    (tsdf_integrator.cc:11,22,29,41)."""
    from voxblox_amd import capi
    from voxblox_amd.integrator import TsdfIntegratorFactory

    class FakeLayer:  # the factory itself never touches the GPU
        pass
    cfg = capi.tsdf_cfg()
    for bad in ("", "octomap", 0, 7):
        with pytest.raises(ValueError):
            TsdfIntegratorFactory.create(bad, cfg, FakeLayer())
    with pytest.raises(ValueError):
        TsdfIntegratorFactory.create("fast", cfg, None)
    assert type(TsdfIntegratorFactory.create("simple", cfg, FakeLayer())).__name__ == "SimpleTsdfIntegrator"
    assert type(TsdfIntegratorFactory.create(2, cfg, FakeLayer())).__name__ == "MergedTsdfIntegrator"
    assert type(TsdfIntegratorFactory.create("fast", cfg, FakeLayer())).__name__ == "FastTsdfIntegrator"


@pytest.mark.parametrize("n,mask", [(0, 0xFFFFFFFF), (1, 0xFFFFFFFF), (13, 0xFFFFFFFF), (14, 0xFFFFFFFF), (30, 0xFFFFFFFF),
                                    (100, 0xFF), (1000, 0), (5000, 0xFFFFFFFF), (31000, 0xFFFFFFFF), (31000, 0x3FF),
                                    (120000, 0xFFFFFFFF)])
def test_unordered_map_order_reconstruction_matches_libstdcxx(n, mask):
    """Host-only self-test (no GPU): the Merged integrator's array replay of libstdc++'s
    unordered_map iteration order (vbx_host_tsdf.hpp) equals the order of the container itself —
    across every rehash threshold up to n, with colliding and with identical hashes."""
    from voxblox_amd import capi
    for seed in (1, 2, 3):
        assert capi.lib().vbx_selftest_unordered_order(n, seed, mask) == 0


@pytest.mark.parametrize("n,span", [(1, 4), (12, 3), (200, 6), (3000, 12), (20000, 40)])
def test_index_set_order_is_the_oracles_container_order(n, span):
    """Host-only (no GPU): the library's stand-in for the reference's IndexSet / HierarchicalIndexMap (reference-order
    addNewRobotPosition: raise_ / open_ pushes and updated_blocks_ come out in the iteration order of an unordered
    container keyed with AnyIndexHash) iterates like the oracle's Layer block map — the same hash in the same libstdc++
    container — after the same sequence of insertions, negative indices and duplicates included."""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    import oracle_py as O
    from voxblox_amd import capi
    rng = np.random.default_rng(n)
    idx = rng.integers(-span, span, size=(n, 3)).astype(np.int32)
    out = np.zeros((n, 3), np.int32)
    k = capi.lib().vbx_selftest_index_set_order(idx.ctypes.data_as(C.POINTER(C.c_int32)), n,
                                                out.ctypes.data_as(C.POINTER(C.c_int32)))
    om = O.OracleMap(0.1, 8)
    z = np.zeros(512, np.float32)
    c = np.zeros((512, 4), np.uint8)
    for i in idx:     # Layer::allocateBlockPtrByIndex in sequence (a second insertion of a key changes nothing)
        om.tsdf_block_set(tuple(int(v) for v in i), z, z, c, 0)
    ref = om.block_indices(0)
    assert k == len(ref) == len({tuple(i) for i in idx.tolist()})
    assert np.array_equal(out[:k], ref)
