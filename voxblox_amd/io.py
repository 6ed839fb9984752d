"""`.voxblox` layer files and voxblox_msgs/Block word streams.

File = one or more layer sections back to back; a section is
    varint32 N (= 1 + #blocks) | varint32 len + LayerProto | (N-1) x (varint32 len + BlockProto)
(protobuf_utils.cc:29-98, layer_inl.h:82-189), proto2 with every field explicitly set:
    LayerProto  1 voxel_size f64 (0x09) | 2 voxels_per_side varint (0x10) | 3 type string (0x1A)
    BlockProto  1 voxels_per_side varint (0x08) | 2 voxel_size f64 (0x11) | 3/4/5 origin_xyz f64
                (0x19/0x21/0x29, the float origin widened) | 6 has_data (0x30) |
                7 voxel_data, UNPACKED: 0x38 + varint32 per word (Block.proto:15)
The words are Block::serializeToIntegers' (block.cc), produced on the GPU by
vbx_blocks_serialize.  No libprotobuf involved: the encoder/decoder below is hand-rolled and
checked byte-for-byte against the python protobuf runtime in tests/test_io_voxblox.py.
Block order in a file is unspecified in the reference (unordered_map order); here ascending
(z,y,x).
"""
import os
import struct

import numpy as np

from . import capi

VOXEL_TYPES = {capi.LAYER_TSDF: "tsdf", capi.LAYER_ESDF: "esdf"}  # core/voxel.h:50-56


def _varint(n):
    n = int(n)
    if n < 0:
        n += 1 << 64  # int32 fields sign-extend to 10 bytes
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    shift = 0
    val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if b < 0x80:
            return val, pos
        shift += 7


def encode_layer_proto(voxel_size, voxels_per_side, type_name):
    """Layer::getProto (layer_inl.h:44-54): voxel_size is the float widened to double."""
    t = type_name.encode()
    return (b"\x09" + struct.pack("<d", float(np.float32(voxel_size))) + b"\x10" + _varint(voxels_per_side)
            + b"\x1a" + _varint(len(t)) + t)


def encode_voxel_data(words):
    """repeated uint32, unpacked: tag 0x38 + varint per word, vectorised."""
    w = np.ascontiguousarray(words, np.uint32).astype(np.uint64).reshape(-1)
    n = w.shape[0]
    out = np.zeros((n, 6), np.uint8)
    out[:, 0] = 0x38
    keep = np.zeros((n, 6), bool)
    keep[:, 0] = True
    rest = w.copy()
    for k in range(5):
        byte = (rest & 0x7F).astype(np.uint8)
        rest = rest >> np.uint64(7)
        more = rest != 0
        out[:, 1 + k] = byte | (more.astype(np.uint8) << 7)
        keep[:, 1 + k] = True if k == 0 else prev_more
        prev_more = more
    return out[keep].tobytes()


def encode_block_proto(voxels_per_side, voxel_size, origin, has_data, words):
    """Block::getProto (block_inl.h:90-109)."""
    o = [float(np.float32(v)) for v in origin]
    head = (b"\x08" + _varint(voxels_per_side) + b"\x11" + struct.pack("<d", float(np.float32(voxel_size)))
            + b"\x19" + struct.pack("<d", o[0]) + b"\x21" + struct.pack("<d", o[1]) + b"\x29" + struct.pack("<d", o[2])
            + b"\x30" + (b"\x01" if has_data else b"\x00"))
    return head + encode_voxel_data(words)


def block_origin(idx, voxel_size, voxels_per_side):
    """getOriginPointFromGridIndex(index, block_size) with block_size = voxel_size * vps in fp32
    (common.h:195-201, layer.h:39)."""
    bs = np.float32(np.float32(voxel_size) * np.float32(voxels_per_side))
    return (np.asarray(idx, np.int32).astype(np.float32) * bs).astype(np.float32)


def block_index_from_origin(origin, voxel_size, voxels_per_side):
    """getGridIndexFromOriginPoint(origin, block_size_inv) (layer_inl.h:199-200, common.h:179-185)."""
    bs = np.float32(np.float32(voxel_size) * np.float32(voxels_per_side))
    inv = np.float32(1.0 / np.float64(bs))
    v = np.asarray(origin, np.float64).astype(np.float32) * inv
    # std::round: half away from zero
    return (np.sign(v) * np.floor(np.abs(v) + np.float32(0.5))).astype(np.int32)


def write_section(f, voxel_size, voxels_per_side, type_name, blocks):
    """blocks: iterable of (index xyz, has_data, words)."""
    blocks = list(blocks)
    f.write(_varint(1 + len(blocks)))
    lp = encode_layer_proto(voxel_size, voxels_per_side, type_name)
    f.write(_varint(len(lp)) + lp)
    for idx, has_data, words in blocks:
        bp = encode_block_proto(voxels_per_side, voxel_size, block_origin(idx, voxel_size, voxels_per_side),
                                has_data, words)
        f.write(_varint(len(bp)) + bp)


def save_layer(gmap, path, layer=capi.LAYER_TSDF, clear_file=True, blocks_to_include=None):
    """io::SaveLayer / Layer::saveToFile / saveSubsetToFile (layer_inl.h:82-158).  With
    clear_file=False the section is appended (multiple layers in one file, :99-106)."""
    if not path:
        raise ValueError("file_path must not be empty")  # CHECK(!file_path.empty())
    idx = gmap.block_indices(layer)
    if blocks_to_include is not None:
        want = {tuple(int(v) for v in b) for b in np.asarray(blocks_to_include).reshape(-1, 3)}
        idx = np.array([i for i in idx if tuple(int(v) for v in i) in want], np.int32).reshape(-1, 3)
    words, has_data = gmap.blocks_serialize(idx, layer) if idx.shape[0] else (np.zeros((0, 0), np.uint32), [])
    with open(path, "wb" if clear_file else "ab") as f:
        write_section(f, gmap.voxel_size, gmap.vps, VOXEL_TYPES[layer],
                      ((idx[i], bool(has_data[i]), words[i]) for i in range(idx.shape[0])))
    return True


def _decode_voxel_data(buf):
    """Vectorised inverse of encode_voxel_data over a bytes region that holds only field 7."""
    b = np.frombuffer(buf, np.uint8)
    if b.shape[0] == 0:
        return np.zeros(0, np.uint32)
    term = np.nonzero(b < 0x80)[0]           # tag bytes and last bytes of varints alternate
    tags, ends = term[0::2], term[1::2]
    if tags.shape[0] != ends.shape[0] or not np.all(b[tags] == 0x38):
        raise ValueError("voxel_data region is not a pure unpacked uint32 stream")
    starts = tags + 1
    lens = ends - starts + 1
    if lens.max() > 5:
        raise ValueError("varint32 longer than 5 bytes")
    vals = np.zeros(starts.shape[0], np.uint64)
    for k in range(5):
        m = lens > k
        vals[m] |= (b[starts[m] + k].astype(np.uint64) & np.uint64(0x7F)) << np.uint64(7 * k)
    return (vals & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def decode_block_proto(buf):
    pos = 0
    out = dict(voxels_per_side=0, voxel_size=0.0, origin=[0.0, 0.0, 0.0], has_data=False)
    n = len(buf)
    while pos < n:
        tag = buf[pos]
        if tag == 0x38:  # everything from the first voxel_data entry on is voxel_data
            out["words"] = _decode_voxel_data(buf[pos:])
            return out
        pos += 1
        if tag == 0x08:
            out["voxels_per_side"], pos = _read_varint(buf, pos)
        elif tag in (0x11, 0x19, 0x21, 0x29):
            v = struct.unpack_from("<d", buf, pos)[0]
            pos += 8
            if tag == 0x11:
                out["voxel_size"] = v
            else:
                out["origin"][(tag - 0x19) // 8] = v
        elif tag == 0x30:
            v, pos = _read_varint(buf, pos)
            out["has_data"] = bool(v)
        else:
            raise ValueError(f"unexpected BlockProto tag 0x{tag:02x}")
    out["words"] = np.zeros(0, np.uint32)
    return out


def decode_layer_proto(buf):
    pos = 0
    out = {}
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        if tag == 0x09:
            out["voxel_size"] = struct.unpack_from("<d", buf, pos)[0]
            pos += 8
        elif tag == 0x10:
            out["voxels_per_side"], pos = _read_varint(buf, pos)
        elif tag == 0x1A:
            ln, pos = _read_varint(buf, pos)
            out["type"] = bytes(buf[pos:pos + ln]).decode()
            pos += ln
        else:
            raise ValueError(f"unexpected LayerProto tag 0x{tag:02x}")
    return out


def read_file(path):
    """All sections of a .voxblox file: [(layer header dict, [block dicts])]."""
    buf = open(path, "rb").read()
    pos = 0
    sections = []
    while pos < len(buf):
        n, pos = _read_varint(buf, pos)
        if n == 0:
            raise ValueError("Empty protobuf file!")  # layer_io_inl.h:159-162
        ln, pos = _read_varint(buf, pos)
        header = decode_layer_proto(buf[pos:pos + ln])
        pos += ln
        blocks = []
        for _ in range(n - 1):
            ln, pos = _read_varint(buf, pos)
            blocks.append(decode_block_proto(buf[pos:pos + ln]))
            pos += ln
        sections.append((header, blocks))
    return sections


def load_layer(path, gmap=None, layer=capi.LAYER_TSDF, multiple_layer_support=False, max_blocks=0, device=0):
    """io::LoadLayer / LoadBlocksFromFile with BlockMergingStrategy::kReplace
    (layer_io_inl.h:15-225): takes the first section whose type matches `layer` (only the
    first section unless multiple_layer_support), checks compatibility (layer_inl.h:232-260),
    uploads the blocks (index = round(origin * block_size_inv), all Update bits set)."""
    if not path:
        raise ValueError("file_path must not be empty")
    if not os.path.exists(path):
        raise IOError(f"Could not open protobuf file to load layer: {path}")
    want = VOXEL_TYPES[layer]
    sections = read_file(path)
    if not multiple_layer_support:
        sections = sections[:1]
    for header, blocks in sections:
        if header.get("type") != want:
            continue
        if gmap is None:
            gmap = capi.Map(np.float32(header["voxel_size"]), header["voxels_per_side"], max_blocks=max_blocks,
                            device=device)
        else:
            ok = (abs(header["voxel_size"] - float(gmap.voxel_size)) < np.finfo(np.float32).eps
                  and header["voxels_per_side"] == gmap.vps)
            if not ok:
                raise ValueError("The blocks from this protobuf are not compatible with this layer!")
        if blocks:
            idx = np.stack([block_index_from_origin(b["origin"], gmap.voxel_size, gmap.vps) for b in blocks])
            words = np.stack([b["words"] for b in blocks])
            hd = np.array([b["has_data"] for b in blocks], np.uint8)
            for b in blocks:
                if b["voxels_per_side"] != gmap.vps:
                    raise ValueError("The blocks from this protobuf are not compatible with this layer!")
            gmap.blocks_deserialize(idx, words, hd, layer)
        return gmap
    raise ValueError(f"no '{want}' layer in {path}")
