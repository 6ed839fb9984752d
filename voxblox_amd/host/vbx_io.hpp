// `.voxblox` layer files for the C++ host shim: io::SaveLayer / io::LoadLayer with the
// reference's signatures (include/voxblox/io/layer_io.h, layer_io_inl.h:131-232;
// Layer::saveToFile / saveSubsetToFile, layer_inl.h:82-189).  The proto2 wire format of
// LayerProto / BlockProto (proto/voxblox/*.proto) is written and parsed by hand — no
// libprotobuf — and the uint32 word stream comes from the GPU (vbx_blocks_serialize =
// Block::serializeToIntegers).  Format notes: SURVEY.md Appendix B; byte-level checks against
// the protobuf runtime: tests/test_io_voxblox.py (python twin of this file: voxblox_amd/io.py).
#ifndef VBX_IO_HPP_
#define VBX_IO_HPP_

#include <cmath>
#include <cstring>
#include <fstream>
#include <limits>
#include <string>
#include <vector>

#include "vbx_integrators.hpp"

namespace vbx_host {
namespace io {
namespace detail {
inline void putVarint(std::string* s, uint64_t v) {
  while (v >= 0x80) { s->push_back(static_cast<char>((v & 0x7F) | 0x80)); v >>= 7; }
  s->push_back(static_cast<char>(v));
}
inline void putDouble(std::string* s, uint8_t tag, double d) {
  s->push_back(static_cast<char>(tag));
  char b[8];
  std::memcpy(b, &d, 8);
  s->append(b, 8);
}
inline bool getVarint(const std::string& s, size_t* pos, uint64_t* v) {
  *v = 0;
  for (int shift = 0; *pos < s.size() && shift < 70; shift += 7) {
    const uint8_t b = static_cast<uint8_t>(s[(*pos)++]);
    *v |= static_cast<uint64_t>(b & 0x7F) << shift;
    if (b < 0x80) return true;
  }
  return false;
}
template <typename V> inline const char* typeName();
template <> inline const char* typeName<TsdfVoxel>() { return "tsdf"; }  // core/voxel.h:50-56
template <> inline const char* typeName<EsdfVoxel>() { return "esdf"; }
template <typename V> inline size_t wordsPerVoxel();
template <> inline size_t wordsPerVoxel<TsdfVoxel>() { return 3; }
template <> inline size_t wordsPerVoxel<EsdfVoxel>() { return 2; }
}  // namespace detail

// Layer::saveSubsetToFile (layer_inl.h:90-158): varint count, LayerProto, then one BlockProto
// per block; clear_file = false appends another layer section to the same file (:99-106).
template <typename VoxelType>
bool SaveLayerSubset(const Layer<VoxelType>& layer, const std::string& file_path, const BlockIndexList& blocks,
                     bool clear_file = true) {
  VBX_CHECK(!file_path.empty(), "file_path");
  std::ofstream out(file_path, std::ios::binary | (clear_file ? std::ios::trunc : std::ios::app));
  if (!out.is_open()) return false;
  const size_t vps = layer.voxels_per_side();
  const size_t wpb = vps * vps * vps * detail::wordsPerVoxel<VoxelType>();
  std::vector<uint32_t> words(blocks.size() * wpb);
  std::vector<uint8_t> has_data(blocks.size() + 1);
  if (!blocks.empty())
    layer.map()->check(vbx_blocks_serialize(layer.map()->ctx(), LayerId<VoxelType>::value, &blocks[0].x,
                                            blocks.size(), words.data(), has_data.data()),
                       "vbx_blocks_serialize");
  std::string buf;
  detail::putVarint(&buf, 1 + blocks.size());
  std::string lp;  // Layer::getProto, layer_inl.h:44-54
  detail::putDouble(&lp, 0x09, static_cast<double>(layer.voxel_size()));
  lp.push_back(0x10); detail::putVarint(&lp, vps);
  const std::string type = detail::typeName<VoxelType>();
  lp.push_back(0x1A); detail::putVarint(&lp, type.size()); lp += type;
  detail::putVarint(&buf, lp.size());
  buf += lp;
  out.write(buf.data(), static_cast<std::streamsize>(buf.size()));
  const float block_size = layer.voxel_size() * static_cast<float>(vps);  // layer.h:39
  for (size_t i = 0; i < blocks.size(); ++i) {
    std::string bp;  // Block::getProto, block_inl.h:90-109
    bp.reserve(64 + wpb * 6);
    bp.push_back(0x08); detail::putVarint(&bp, vps);
    detail::putDouble(&bp, 0x11, static_cast<double>(layer.voxel_size()));
    detail::putDouble(&bp, 0x19, static_cast<double>(static_cast<float>(blocks[i].x) * block_size));
    detail::putDouble(&bp, 0x21, static_cast<double>(static_cast<float>(blocks[i].y) * block_size));
    detail::putDouble(&bp, 0x29, static_cast<double>(static_cast<float>(blocks[i].z) * block_size));
    bp.push_back(0x30); bp.push_back(has_data[i] ? 1 : 0);
    for (size_t w = 0; w < wpb; ++w) {  // repeated uint32, NOT packed (Block.proto:15)
      bp.push_back(0x38);
      detail::putVarint(&bp, words[i * wpb + w]);
    }
    std::string head;
    detail::putVarint(&head, bp.size());
    out.write(head.data(), static_cast<std::streamsize>(head.size()));
    out.write(bp.data(), static_cast<std::streamsize>(bp.size()));
  }
  return out.good();
}

template <typename VoxelType>
bool SaveLayer(const Layer<VoxelType>& layer, const std::string& file_path, bool clear_file = true) {
  BlockIndexList blocks;
  layer.getAllAllocatedBlocks(&blocks);
  return SaveLayerSubset(layer, file_path, blocks, clear_file);
}

// io::LoadBlocksFromFile with BlockMergingStrategy::kReplace (layer_io_inl.h:15-92): loads the
// first section of matching voxel type (only the first section unless multiple_layer_support)
// into an existing, compatible layer.
template <typename VoxelType>
bool LoadBlocksFromFile(const std::string& file_path, bool multiple_layer_support, Layer<VoxelType>* layer) {
  VBX_CHECK(layer != nullptr, "layer_ptr");
  VBX_CHECK(!file_path.empty(), "file_path");
  std::ifstream in(file_path, std::ios::binary);
  if (!in.is_open()) return false;
  const std::string s((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  size_t pos = 0;
  const size_t vps = layer->voxels_per_side();
  const size_t wpb = vps * vps * vps * detail::wordsPerVoxel<VoxelType>();
  const float block_size = layer->voxel_size() * static_cast<float>(vps);
  const float block_size_inv = static_cast<float>(1.0 / static_cast<double>(block_size));
  bool first = true;
  while (pos < s.size()) {
    uint64_t n = 0, len = 0;
    if (!detail::getVarint(s, &pos, &n) || n == 0) return false;
    if (!detail::getVarint(s, &pos, &len) || pos + len > s.size()) return false;
    double voxel_size = 0; uint64_t file_vps = 0; std::string type;
    for (size_t p = pos, e = pos + len; p < e;) {
      const uint8_t tag = static_cast<uint8_t>(s[p++]);
      if (tag == 0x09) { std::memcpy(&voxel_size, &s[p], 8); p += 8; }
      else if (tag == 0x10) { if (!detail::getVarint(s, &p, &file_vps)) return false; }
      else if (tag == 0x1A) { uint64_t l; if (!detail::getVarint(s, &p, &l)) return false; type = s.substr(p, l); p += l; }
      else return false;
    }
    pos += len;
    const bool wanted = (type == detail::typeName<VoxelType>()) && (first || multiple_layer_support);
    if (wanted) {  // Layer::isCompatible, layer_inl.h:232-260
      if (!(std::fabs(voxel_size - layer->voxel_size()) < std::numeric_limits<float>::epsilon()) || file_vps != vps)
        return false;
    }
    BlockIndexList idx;
    std::vector<uint32_t> words;
    std::vector<uint8_t> has_data;
    for (uint64_t b = 0; b + 1 < n; ++b) {
      if (!detail::getVarint(s, &pos, &len) || pos + len > s.size()) return false;
      const size_t e = pos + len;
      double ox = 0, oy = 0, oz = 0; uint64_t hd = 0, bvps = 0; size_t nw = 0;
      if (wanted) words.resize(words.size() + wpb);
      for (size_t p = pos; p < e;) {
        const uint8_t tag = static_cast<uint8_t>(s[p++]);
        uint64_t v;
        if (tag == 0x08) { if (!detail::getVarint(s, &p, &bvps)) return false; }
        else if (tag == 0x11) { p += 8; }
        else if (tag == 0x19) { std::memcpy(&ox, &s[p], 8); p += 8; }
        else if (tag == 0x21) { std::memcpy(&oy, &s[p], 8); p += 8; }
        else if (tag == 0x29) { std::memcpy(&oz, &s[p], 8); p += 8; }
        else if (tag == 0x30) { if (!detail::getVarint(s, &p, &hd)) return false; }
        else if (tag == 0x38) {
          if (!detail::getVarint(s, &p, &v)) return false;
          if (wanted) { if (nw >= wpb) return false; words[words.size() - wpb + nw] = static_cast<uint32_t>(v); }
          ++nw;
        } else return false;
      }
      pos = e;
      if (wanted) {
        if (nw != wpb || bvps != vps) return false;  // CHECK_EQ in deserializeFromIntegers / isCompatible
        // getGridIndexFromOriginPoint(origin, block_size_inv), layer_inl.h:199-200
        idx.push_back({static_cast<int32_t>(std::round(static_cast<float>(ox) * block_size_inv)),
                       static_cast<int32_t>(std::round(static_cast<float>(oy) * block_size_inv)),
                       static_cast<int32_t>(std::round(static_cast<float>(oz) * block_size_inv))});
        has_data.push_back(hd ? 1 : 0);
      }
    }
    if (wanted) {
      if (!idx.empty())
        layer->map()->check(vbx_blocks_deserialize(layer->map()->ctx(), LayerId<VoxelType>::value, &idx[0].x,
                                                   idx.size(), words.data(), has_data.data()),
                            "vbx_blocks_deserialize");
      return true;
    }
    first = false;
    if (!multiple_layer_support) return false;
  }
  return false;
}

}  // namespace io
}  // namespace vbx_host

#endif  // VBX_IO_HPP_
