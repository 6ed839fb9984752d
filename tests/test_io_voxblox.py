"""Layer serialization.  CPU part: the hand-rolled proto2
encoder/decoder against the python protobuf runtime (descriptors built from
proto/voxblox/{Block,Layer}.proto's field lists) byte for byte, and the oracle's word packing
against the reference build (incl. the sign-extension of negative ESDF parents).  GPU part:
vbx_blocks_serialize / deserialize and .voxblox round trips through the C-ABI."""
import ctypes as C
import os

import numpy as np
import pytest

from voxblox_amd import io as vio
from voxblox_amd import scenes


def _pb_classes():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = "voxblox_test.proto"
    fd.package = "voxblox"
    fd.syntax = "proto2"
    F = descriptor_pb2.FieldDescriptorProto
    blk = fd.message_type.add()
    blk.name = "BlockProto"
    for num, name, typ, label in ((1, "voxels_per_side", F.TYPE_INT32, F.LABEL_OPTIONAL),
                                  (2, "voxel_size", F.TYPE_DOUBLE, F.LABEL_OPTIONAL),
                                  (3, "origin_x", F.TYPE_DOUBLE, F.LABEL_OPTIONAL),
                                  (4, "origin_y", F.TYPE_DOUBLE, F.LABEL_OPTIONAL),
                                  (5, "origin_z", F.TYPE_DOUBLE, F.LABEL_OPTIONAL),
                                  (6, "has_data", F.TYPE_BOOL, F.LABEL_OPTIONAL),
                                  (7, "voxel_data", F.TYPE_UINT32, F.LABEL_REPEATED)):
        f = blk.field.add()
        f.name, f.number, f.type, f.label = name, num, typ, label
    lay = fd.message_type.add()
    lay.name = "LayerProto"
    for num, name, typ in ((1, "voxel_size", F.TYPE_DOUBLE), (2, "voxels_per_side", F.TYPE_UINT32),
                           (3, "type", F.TYPE_STRING)):
        f = lay.field.add()
        f.name, f.number, f.type, f.label = name, num, typ, F.LABEL_OPTIONAL
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, "GetMessageClass", None)
    if get is None:
        fac = message_factory.MessageFactory(pool)
        get = fac.GetPrototype
    return (get(pool.FindMessageTypeByName("voxblox.BlockProto")),
            get(pool.FindMessageTypeByName("voxblox.LayerProto")))


def test_proto_encoding_matches_protobuf_runtime():
    BlockProto, LayerProto = _pb_classes()
    rng = np.random.RandomState(0)
    words = np.concatenate([rng.randint(0, 2 ** 32, 3000, dtype=np.uint64).astype(np.uint32),
                            np.array([0, 1, 127, 128, 16383, 16384, 2 ** 21 - 1, 2 ** 21, 2 ** 28 - 1, 2 ** 28,
                                      2 ** 32 - 1], np.uint32)])
    for has_data in (False, True):
        origin = vio.block_origin((-3, 2, 7), 0.05, 16)
        mine = vio.encode_block_proto(16, 0.05, origin, has_data, words)
        m = BlockProto()
        m.voxels_per_side = 16
        m.voxel_size = float(np.float32(0.05))
        m.origin_x, m.origin_y, m.origin_z = (float(v) for v in origin)
        m.has_data = has_data
        m.voxel_data.extend(int(w) for w in words)
        assert mine == m.SerializeToString()
        back = vio.decode_block_proto(mine)
        assert np.array_equal(back["words"], words) and back["has_data"] == has_data
        assert back["voxels_per_side"] == 16 and back["origin"] == [float(v) for v in origin]
        p = BlockProto()
        p.ParseFromString(mine)
        assert list(p.voxel_data) == [int(w) for w in words]
    for t in ("tsdf", "esdf"):
        mine = vio.encode_layer_proto(0.2, 16, t)
        m = LayerProto()
        m.voxel_size = float(np.float32(0.2))
        m.voxels_per_side = 16
        m.type = t
        assert mine == m.SerializeToString()
        assert vio.decode_layer_proto(mine) == dict(voxel_size=float(np.float32(0.2)), voxels_per_side=16, type=t)


def test_block_origin_index_round_trip():
    for vs in (0.02, 0.05, 0.1, 0.2):
        for idx in ((0, 0, 0), (-1, -1, -1), (5, -7, 13), (-50, 50, 1)):
            o = vio.block_origin(idx, vs, 16)
            assert tuple(vio.block_index_from_origin(o.astype(np.float64), vs, 16)) == idx


def _random_esdf_block(rng):
    n = 4096
    d = rng.uniform(-2, 2, n).astype(np.float32)
    fl = rng.randint(0, 16, n).astype(np.uint8)
    par = rng.randint(-1, 2, (n, 3)).astype(np.int32)
    par[:50] = rng.randint(-200, 200, (50, 3))      # exercises the int8 clamp (block.cc:27-32)
    return d, fl, par


def test_oracle_word_packing_vs_reference_build(oracle):
    """serializeToIntegers / deserializeFromIntegers restatement == the reference's block.cc,
    including the sign-extension of negative parent components."""
    if not oracle.ref_available():
        pytest.skip("oracle/_ref not available")
    rng = np.random.RandomState(1)
    d, fl, par = _random_esdf_block(rng)
    td = rng.uniform(-0.2, 0.2, 4096).astype(np.float32)
    tw = rng.uniform(0, 100, 4096).astype(np.float32)
    tc = rng.randint(0, 256, (4096, 4)).astype(np.uint8)
    got = {}
    for name, L in (("oracle", oracle.lib()), ("ref", oracle.ref_lib())):
        m = oracle.OracleMap(0.05, 16, L=L)
        m.esdf_block_set((1, -2, 3), d, fl, par)
        m.tsdf_block_set((1, -2, 3), td, tw, tc)
        we, wt = m.block_serialize((1, -2, 3), 1), m.block_serialize((1, -2, 3), 0)
        m2 = oracle.OracleMap(0.05, 16, L=L)
        assert m2.block_deserialize((4, 4, 4), we, 1) and m2.block_deserialize((4, 4, 4), wt, 0)
        assert not m2.block_deserialize((5, 5, 5), wt[:-1], 0)
        got[name] = (we, wt, m2.esdf_block((4, 4, 4)), m2.tsdf_block((4, 4, 4)))
    assert np.array_equal(got["oracle"][0], got["ref"][0]) and np.array_equal(got["oracle"][1], got["ref"][1])
    for a, b in zip(got["oracle"][2], got["ref"][2]):
        assert np.array_equal(a, b)
    for a, b in zip(got["oracle"][3], got["ref"][3]):
        assert np.array_equal(a, b)
    # the documented quirk: parent (1,-1,0) -> 0xFFFF0000 | flags
    m = oracle.OracleMap(0.05, 16)
    par2 = np.zeros((4096, 3), np.int32); par2[0] = (1, -1, 0)
    m.esdf_block_set((0, 0, 0), np.zeros(4096, np.float32), np.zeros(4096, np.uint8), par2)
    assert m.block_serialize((0, 0, 0), 1)[1] == 0xFFFF0000


@pytest.mark.gpu
@pytest.mark.parametrize("reference_order", [1, 0])
def test_gpu_words_and_file_round_trip(oracle, tmp_path, reference_order):
    """reference_order = 1 (the C-ABI's default): the ESDF word stream — distance, flags AND parents (block.cc:112-137) —
    equals the unswitched reference's word for word; 0: the order-free mode against the oracle's order-free switch, where
    parents may differ on ties and are masked."""
    from voxblox_amd import capi
    voxel = 0.1
    gm = capi.Map(voxel, 16, max_blocks=2048)
    om = oracle.OracleMap(voxel, 16)
    oi = om.tsdf_integrator("simple", oracle.tsdf_cfg(default_truncation_distance=0.4, integrator_threads=1))
    if reference_order:
        oe = om.esdf_integrator(oracle.esdf_cfg(min_distance_m=0.2))
    else:
        oe = om.esdf_integrator(oracle.esdf_cfg(min_distance_m=0.2, min_diff_m=0.0, oracle_orderfree_sign_mismatch=1))
    for k in range(2):
        pose, pts, col = scenes.room_frame(9 * k, 100, f=40.0, width=80, height=60)
        gm.integrate(capi.TSDF_SIMPLE, capi.tsdf_cfg(default_truncation_distance=0.4), pose[0], pose[1], pts, col)
        oi.integrate(pose[0], pose[1], pts, col)
    if reference_order:
        gm.esdf_update(capi.esdf_cfg(reference_order=1, min_distance_m=0.2), batch=True)
        assert gm.counters()["esdf_order_inexact"] == 0
    else:
        gm.esdf_update(capi.esdf_cfg(reference_order=0, min_distance_m=0.2, min_diff_m=0.0), batch=True)
    oe.update_from_tsdf_layer_batch()
    # 1. word streams bit-identical to the oracle's (== reference build's) for TSDF; for ESDF
    #    distance + flag bits identical, parents may differ on ties -> compare after masking
    idx = gm.block_indices()
    wt, hd = gm.blocks_serialize(idx, capi.LAYER_TSDF)
    assert not hd.any()                      # integrators never set has_data
    for i, b in enumerate(idx):
        assert np.array_equal(wt[i], om.block_serialize(b, 0))
    we, _ = gm.blocks_serialize(idx, capi.LAYER_ESDF)
    for i, b in enumerate(idx):
        ow = om.block_serialize(b, 1)
        if reference_order:
            assert np.array_equal(we[i], ow), f"ESDF words (distance, parents, flags) differ in block {tuple(b)}"
        else:
            assert np.array_equal(we[i][0::2], ow[0::2])
            assert np.array_equal(we[i][1::2] & 0xF, ow[1::2] & 0xF)
    # 2. .voxblox file: TSDF + appended ESDF section, reloaded into a fresh map
    path = str(tmp_path / "map.voxblox")
    vio.save_layer(gm, path, capi.LAYER_TSDF, clear_file=True)
    vio.save_layer(gm, path, capi.LAYER_ESDF, clear_file=False)
    sections = vio.read_file(path)
    assert [s[0]["type"] for s in sections] == ["tsdf", "esdf"]
    assert all(len(s[1]) == len(idx) for s in sections)
    g2 = vio.load_layer(path, layer=capi.LAYER_TSDF, max_blocks=2048)
    vio.load_layer(path, gmap=g2, layer=capi.LAYER_ESDF, multiple_layer_support=True)
    a, b = gm.tsdf_dict(), g2.tsdf_dict()
    assert set(a) == set(b)
    for k in a:
        assert np.array_equal(a[k][0].view(np.uint32), b[k][0].view(np.uint32))
        assert np.array_equal(a[k][1].view(np.uint32), b[k][1].view(np.uint32))
        assert np.array_equal(a[k][2], b[k][2]) and b[k][3] == 7   # addBlockFromProto sets all bits
    for i in idx:
        va, _, _ = gm.block_download(i, capi.LAYER_ESDF)
        vb, ub, _ = g2.block_download(i, capi.LAYER_ESDF)
        assert np.array_equal(va["distance"].view(np.uint32), vb["distance"].view(np.uint32))
        for f in ("observed", "hallucinated", "in_queue", "fixed"):
            assert np.array_equal(va[f], vb[f])
        # parents survive unless a component is negative (the reference writer's bug)
        nonneg = (va["parent"] >= 0).all(axis=1)
        assert np.array_equal(va["parent"][nonneg], vb["parent"][nonneg])
        assert ub == 7
    # 3. the oracle reads the GPU's words back to identical voxels
    o2 = oracle.OracleMap(voxel, 16)
    for i, b in enumerate(idx):
        assert o2.block_deserialize(b, wt[i], 0)
        d, w, c, _ = o2.tsdf_block(b)
        key = tuple(int(v) for v in b)
        assert np.array_equal(d, a[key][0]) and np.array_equal(w, a[key][1]) and np.array_equal(c, a[key][2])
    # 4. wrong geometry is refused like Layer::isCompatible (layer_inl.h:232-260)
    g3 = capi.Map(0.2, 16, max_blocks=64)
    with pytest.raises(ValueError):
        vio.load_layer(path, gmap=g3)
    # 5. subset save (saveSubsetToFile)
    vio.save_layer(gm, str(tmp_path / "sub.voxblox"), blocks_to_include=idx[:3])
    assert len(vio.read_file(str(tmp_path / "sub.voxblox"))[0][1]) == 3
