// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Dependency-free CPU restatement of the voxblox map containers and index math
// that sit on the TSDF/ESDF integration hot path.  Only tests/, bench.py's
// cpu_baseline leg and __graft_entry__.smoke() may use anything under oracle/;
// the product path (voxblox_amd/csrc) never links or calls it.
//
// Parity status: the reference cannot be compiled in this image (Eigen, glog,
// minkindr, protobuf absent), so this restatement is checked against the
// reference's own known-answer tests (tests/test_oracle_known_answers.py) and,
// when /root/reference is present, against the reference sources compiled over
// dependency shims (oracle/_ref, see oracle/Makefile).  Third-party arithmetic
// (Eigen reductions, minkindr transform) is restated from their published
// algorithms; each site says so.
//
// Every function cites the reference file:line it follows (paths relative to
// /root/reference/voxblox).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <unordered_map>
#include <vector>

namespace orc {

// include/voxblox/core/common.h:41-43 — scalar types.
using FloatingPoint = float;
using IndexElement = int32_t;
using LongIndexElement = int64_t;

// common.h:144-145
constexpr FloatingPoint kEpsilon = 1e-6;
constexpr float kFloatEpsilon = 1e-6;

// ---------------------------------------------------------------------------
// 3-vectors.  Eigen::Matrix<float,3,1> stand-in with explicit scalar op order.
// Eigen's non-vectorised 3-element reductions associate as c0 + (c1 + c2)
// (redux_novec_unroller halves the range) — restated from Eigen 3.3.
// ---------------------------------------------------------------------------
struct Vec3f {
  float x, y, z;
  float& operator[](int i) { return (&x)[i]; }
  float operator[](int i) const { return (&x)[i]; }
};
inline Vec3f operator+(const Vec3f& a, const Vec3f& b) {
  return {a.x + b.x, a.y + b.y, a.z + b.z};
}
inline Vec3f operator-(const Vec3f& a, const Vec3f& b) {
  return {a.x - b.x, a.y - b.y, a.z - b.z};
}
inline Vec3f operator*(const Vec3f& a, float s) {
  return {a.x * s, a.y * s, a.z * s};
}
inline Vec3f operator/(const Vec3f& a, float s) {
  return {a.x / s, a.y / s, a.z / s};
}
inline float sqnorm(const Vec3f& a) {
  return a.x * a.x + (a.y * a.y + a.z * a.z);
}
inline float norm(const Vec3f& a) { return std::sqrt(sqnorm(a)); }
inline float dot(const Vec3f& a, const Vec3f& b) {
  return a.x * b.x + (a.y * b.y + a.z * b.z);
}
inline Vec3f cross(const Vec3f& a, const Vec3f& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// Eigen 3.3 MatrixBase::normalized(): z = squaredNorm; z > 0 ? v / sqrt(z) : v.
inline Vec3f normalized(const Vec3f& a) {
  const float z = sqnorm(a);
  if (z > 0.0f) return a / std::sqrt(z);
  return a;
}

struct Idx3 {  // AnyIndex / BlockIndex / VoxelIndex (common.h:48-51)
  int32_t x, y, z;
  bool operator==(const Idx3& o) const { return x == o.x && y == o.y && z == o.z; }
  bool operator!=(const Idx3& o) const { return !(*this == o); }
  int32_t& operator[](int i) { return (&x)[i]; }
  int32_t operator[](int i) const { return (&x)[i]; }
};
struct LIdx3 {  // LongIndex / GlobalIndex (common.h:53-54)
  int64_t x, y, z;
  bool operator==(const LIdx3& o) const { return x == o.x && y == o.y && z == o.z; }
  bool operator!=(const LIdx3& o) const { return !(*this == o); }
  int64_t& operator[](int i) { return (&x)[i]; }
  int64_t operator[](int i) const { return (&x)[i]; }
};

// common.h:94-125 — Color and blendTwoColors.
struct Color {
  uint8_t r = 0, g = 0, b = 0, a = 0;
};
inline Color blendTwoColors(const Color& c1, float w1, const Color& c2, float w2) {
  const float total = w1 + w2;
  w1 /= total;
  w2 /= total;
  Color o;
  o.r = static_cast<uint8_t>(std::round(c1.r * w1 + c2.r * w2));
  o.g = static_cast<uint8_t>(std::round(c1.g * w1 + c2.g * w2));
  o.b = static_cast<uint8_t>(std::round(c1.b * w1 + c2.b * w2));
  o.a = static_cast<uint8_t>(std::round(c1.a * w1 + c2.a * w2));
  return o;
}

// common.h:248 — signum.
inline int signum(float x) { return (x == 0) ? 0 : x < 0 ? -1 : 1; }

// ---------------------------------------------------------------------------
// Grid <-> point conversions, common.h:153-243.
// ---------------------------------------------------------------------------
// common.h:153-159 — floor(p * inv + 1e-6f) per axis, all in fp32.
inline LIdx3 gridIndexFromPointL(const Vec3f& p, float inv) {
  return {static_cast<int64_t>(std::floor(p.x * inv + kEpsilon)),
          static_cast<int64_t>(std::floor(p.y * inv + kEpsilon)),
          static_cast<int64_t>(std::floor(p.z * inv + kEpsilon))};
}
inline Idx3 gridIndexFromPointI(const Vec3f& p, float inv) {
  return {static_cast<int32_t>(std::floor(p.x * inv + kEpsilon)),
          static_cast<int32_t>(std::floor(p.y * inv + kEpsilon)),
          static_cast<int32_t>(std::floor(p.z * inv + kEpsilon))};
}
// common.h:166-171 — pre-scaled variant.
inline LIdx3 gridIndexFromScaledPointL(const Vec3f& p) {
  return {static_cast<int64_t>(std::floor(p.x + kEpsilon)),
          static_cast<int64_t>(std::floor(p.y + kEpsilon)),
          static_cast<int64_t>(std::floor(p.z + kEpsilon))};
}
// common.h:179-185 — round(p * inv).
inline Idx3 gridIndexFromOriginPoint(const Vec3f& p, float inv) {
  return {static_cast<int32_t>(std::round(p.x * inv)),
          static_cast<int32_t>(std::round(p.y * inv)),
          static_cast<int32_t>(std::round(p.z * inv))};
}
// common.h:187-193 — the "+ 0.5" literal is a double, so the sum and the
// product are evaluated in double and rounded to float once.
template <typename I>
inline Vec3f centerPointFromGridIndex(const I& idx, float grid_size) {
  return {static_cast<float>((static_cast<float>(idx.x) + 0.5) * grid_size),
          static_cast<float>((static_cast<float>(idx.y) + 0.5) * grid_size),
          static_cast<float>((static_cast<float>(idx.z) + 0.5) * grid_size)};
}
// common.h:195-201
template <typename I>
inline Vec3f originPointFromGridIndex(const I& idx, float grid_size) {
  return {static_cast<float>(idx.x) * grid_size,
          static_cast<float>(idx.y) * grid_size,
          static_cast<float>(idx.z) * grid_size};
}
// common.h:208-213
inline LIdx3 globalVoxelIndexFromBlockAndVoxelIndex(const Idx3& b, const Idx3& v,
                                                    int vps) {
  return {static_cast<int64_t>(b.x) * vps + v.x,
          static_cast<int64_t>(b.y) * vps + v.y,
          static_cast<int64_t>(b.z) * vps + v.z};
}
// common.h:215-224 — floor((float)g * vps_inv) in fp32.
inline Idx3 blockIndexFromGlobalVoxelIndex(const LIdx3& g, float vps_inv) {
  return {static_cast<int32_t>(std::floor(static_cast<float>(g.x) * vps_inv)),
          static_cast<int32_t>(std::floor(static_cast<float>(g.y) * vps_inv)),
          static_cast<int32_t>(std::floor(static_cast<float>(g.z) * vps_inv))};
}
// common.h:233-243 — (g + INT_MIN) & (vps - 1).  `offset` is int(1<<31), which
// is INT_MIN; added to an int64 it sign-extends.
inline Idx3 localFromGlobalVoxelIndex(const LIdx3& g, int vps) {
  const int64_t offset = static_cast<int64_t>(INT32_MIN);
  return {static_cast<int32_t>((g.x + offset) & (vps - 1)),
          static_cast<int32_t>((g.y + offset) & (vps - 1)),
          static_cast<int32_t>((g.z + offset) & (vps - 1))};
}

// ---------------------------------------------------------------------------
// Hashes, core/block_hash.h:20-31 and :54-64.  The sum is formed in size_t
// (wraps mod 2^64), truncated to 32 bits, widened again.
// ---------------------------------------------------------------------------
inline size_t anyIndexHash(const Idx3& i) {
  constexpr size_t sl = 17191, sl2 = sl * sl;
  return static_cast<unsigned int>(static_cast<size_t>(static_cast<int64_t>(i.x)) +
                                   static_cast<size_t>(static_cast<int64_t>(i.y)) * sl +
                                   static_cast<size_t>(static_cast<int64_t>(i.z)) * sl2);
}
inline size_t longIndexHash(const LIdx3& i) {
  constexpr size_t sl = 17191, sl2 = sl * sl;
  return static_cast<unsigned int>(static_cast<size_t>(i.x) +
                                   static_cast<size_t>(i.y) * sl +
                                   static_cast<size_t>(i.z) * sl2);
}
struct AnyIndexHasher {
  size_t operator()(const Idx3& i) const { return anyIndexHash(i); }
};
struct LongIndexHasher {
  size_t operator()(const LIdx3& i) const { return longIndexHash(i); }
};

// ---------------------------------------------------------------------------
// Voxels, core/voxel.h:12-37.
// ---------------------------------------------------------------------------
struct TsdfVoxel {
  float distance = 0.0f;
  float weight = 0.0f;
  Color color;
};
struct EsdfVoxel {
  float distance = 0.0f;
  bool observed = false;
  bool hallucinated = false;
  bool in_queue = false;
  bool fixed = false;
  Idx3 parent{0, 0, 0};
};

// core/block.h:15-18 — Update::Status.
enum UpdateBit { kMap = 0, kMesh = 1, kEsdf = 2, kCount = 3 };

// ---------------------------------------------------------------------------
// Block<V>, core/block.h:23-215, core/block_inl.h:13-54.
// ---------------------------------------------------------------------------
template <typename V>
struct Block {
  Block(size_t vps, float voxel_size, const Vec3f& origin)
      : has_data(false), voxels_per_side(vps), voxel_size(voxel_size),
        origin(origin), updated(0) {
    num_voxels = vps * vps * vps;
    voxel_size_inv = 1.0 / voxel_size;   // block.h:37 (double div -> float)
    block_size = vps * voxel_size;       // block.h:38 (size_t -> float mul)
    block_size_inv = 1.0 / block_size;   // block.h:39
    voxels.reset(new V[num_voxels]);
  }
  // block_inl.h:13-27
  size_t linearIndex(const Idx3& v) const {
    return static_cast<size_t>(v.x + voxels_per_side * (v.y + v.z * voxels_per_side));
  }
  // block_inl.h:43-54
  Idx3 voxelIndexFromLinear(size_t lin) const {
    const int vps = static_cast<int>(voxels_per_side);
    int rem = static_cast<int>(lin);
    Idx3 r;
    r.z = rem / (vps * vps);
    rem = rem % (vps * vps);
    r.y = rem / vps;
    r.x = rem % vps;
    return r;
  }
  bool isValidVoxelIndex(const Idx3& v) const {
    const int vps = static_cast<int>(voxels_per_side);
    return v.x >= 0 && v.x < vps && v.y >= 0 && v.y < vps && v.z >= 0 && v.z < vps;
  }
  // block.h:157-159
  Idx3 block_index() const { return gridIndexFromOriginPoint(origin, block_size_inv); }

  std::unique_ptr<V[]> voxels;
  size_t num_voxels;
  bool has_data;
  size_t voxels_per_side;
  float voxel_size;
  Vec3f origin;
  float voxel_size_inv, block_size, block_size_inv;
  uint8_t updated;  // bitset<3>
};

// ---------------------------------------------------------------------------
// Block <-> uint32 words, src/core/block.cc.  TSDF: 3 words per voxel
// {bits(distance), bits(weight), r<<24|g<<16|b<<8|a} (:160-183, :66-90).  ESDF: 2 words
// {bits(distance), parent<<8 | flags} (:204-234, :112-137) where serializeDirection (:8-40)
// shifts each int8 AFTER promotion to int and before the uint32 cast, so a negative component
// sign-extends over all higher bytes — reproduced for byte compatibility.
// ---------------------------------------------------------------------------
inline void serializeBlock(const Block<TsdfVoxel>& b, std::vector<uint32_t>* data) {
  data->clear();
  data->reserve(b.num_voxels * 3);
  for (size_t i = 0; i < b.num_voxels; ++i) {
    const TsdfVoxel& v = b.voxels[i];
    uint32_t w1, w2;
    std::memcpy(&w1, &v.distance, 4);
    std::memcpy(&w2, &v.weight, 4);
    data->push_back(w1);
    data->push_back(w2);
    data->push_back(static_cast<uint32_t>(v.color.a) | (static_cast<uint32_t>(v.color.b) << 8) |
                    (static_cast<uint32_t>(v.color.g) << 16) | (static_cast<uint32_t>(v.color.r) << 24));
  }
}
inline bool deserializeBlock(const std::vector<uint32_t>& data, Block<TsdfVoxel>* b) {
  if (data.size() != b->num_voxels * 3) return false;  // CHECK_EQ, block.cc:70
  for (size_t i = 0, d = 0; i < b->num_voxels; ++i, d += 3) {
    TsdfVoxel& v = b->voxels[i];
    std::memcpy(&v.distance, &data[d], 4);
    std::memcpy(&v.weight, &data[d + 1], 4);
    const uint32_t c = data[d + 2];
    v.color.r = static_cast<uint8_t>(c >> 24);
    v.color.g = static_cast<uint8_t>((c & 0x00FF0000) >> 16);
    v.color.b = static_cast<uint8_t>((c & 0x0000FF00) >> 8);
    v.color.a = static_cast<uint8_t>(c & 0x000000FF);
  }
  return true;
}
inline uint32_t serializeDirection(const Idx3& p) {
  const int8_t px = static_cast<int8_t>(std::min<int>(INT8_MAX, std::max<int>(p.x, INT8_MIN)));
  const int8_t py = static_cast<int8_t>(std::min<int>(INT8_MAX, std::max<int>(p.y, INT8_MIN)));
  const int8_t pz = static_cast<int8_t>(std::min<int>(INT8_MAX, std::max<int>(p.z, INT8_MIN)));
  uint32_t data = 0;
  // static_cast<int8_t>(x) << 24 is an int shift of the promoted (sign-extended) value
  data |= static_cast<uint32_t>(static_cast<int64_t>(px) << 24);
  data |= static_cast<uint32_t>(static_cast<int64_t>(py) << 16);
  data |= static_cast<uint32_t>(static_cast<int64_t>(pz) << 8);
  return data;
}
inline void serializeBlock(const Block<EsdfVoxel>& b, std::vector<uint32_t>* data) {
  data->clear();
  data->reserve(b.num_voxels * 2);
  for (size_t i = 0; i < b.num_voxels; ++i) {
    const EsdfVoxel& v = b.voxels[i];
    uint32_t w1;
    std::memcpy(&w1, &v.distance, 4);
    data->push_back(w1);
    uint32_t w2 = serializeDirection(v.parent);
    uint8_t flags = 0;
    flags |= v.observed ? 1 : 0;
    flags |= v.hallucinated ? 2 : 0;
    flags |= v.in_queue ? 4 : 0;
    flags |= v.fixed ? 8 : 0;
    w2 |= static_cast<uint32_t>(flags) & 0xFF;
    data->push_back(w2);
  }
}
inline bool deserializeBlock(const std::vector<uint32_t>& data, Block<EsdfVoxel>* b) {
  if (data.size() != b->num_voxels * 2) return false;
  for (size_t i = 0, d = 0; i < b->num_voxels; ++i, d += 2) {
    EsdfVoxel& v = b->voxels[i];
    std::memcpy(&v.distance, &data[d], 4);
    const uint32_t w = data[d + 1];
    v.observed = (w & 1) != 0;
    v.hallucinated = (w & 2) != 0;
    v.in_queue = (w & 4) != 0;
    v.fixed = (w & 8) != 0;
    v.parent = {static_cast<int8_t>((w >> 24) & 0xFF), static_cast<int8_t>((w >> 16) & 0xFF),
                static_cast<int8_t>((w >> 8) & 0xFF)};
  }
  return true;
}

// ---------------------------------------------------------------------------
// Layer<V>, core/layer.h:24-296.  std::unordered_map with the reference's hash
// so that iteration order under this libstdc++ equals what the reference
// would produce for the same insertion sequence.
// ---------------------------------------------------------------------------
template <typename V>
struct Layer {
  using BlockT = Block<V>;
  using BlockPtr = std::shared_ptr<BlockT>;
  using BlockMap = std::unordered_map<Idx3, BlockPtr, AnyIndexHasher>;

  Layer(float voxel_size, size_t vps) : voxel_size(voxel_size), voxels_per_side(vps) {
    voxel_size_inv = 1.0 / voxel_size;            // layer.h:37
    block_size = voxel_size * voxels_per_side;    // layer.h:39
    block_size_inv = 1.0 / block_size;            // layer.h:41
    voxels_per_side_inv = 1.0f / static_cast<float>(voxels_per_side);  // :43
  }
  BlockPtr getBlockPtrByIndex(const Idx3& i) const {
    auto it = block_map.find(i);
    return it == block_map.end() ? BlockPtr() : it->second;
  }
  // layer.h:133-145
  BlockPtr allocateNewBlock(const Idx3& i) {
    auto st = block_map.emplace(
        i, std::make_shared<BlockT>(voxels_per_side, voxel_size,
                                    originPointFromGridIndex(i, block_size)));
    return st.first->second;
  }
  // layer.h:95-103
  BlockPtr allocateBlockPtrByIndex(const Idx3& i) {
    auto it = block_map.find(i);
    return it != block_map.end() ? it->second : allocateNewBlock(i);
  }
  bool hasBlock(const Idx3& i) const { return block_map.count(i) > 0; }
  // layer.h:228-239
  V* getVoxelPtrByGlobalIndex(const LIdx3& g) {
    const Idx3 b = blockIndexFromGlobalVoxelIndex(g, voxels_per_side_inv);
    auto it = block_map.find(b);
    if (it == block_map.end()) return nullptr;
    const Idx3 l = localFromGlobalVoxelIndex(g, static_cast<int>(voxels_per_side));
    return &it->second->voxels[it->second->linearIndex(l)];
  }
  // layer.h:194-203
  void getAllUpdatedBlocks(int bit, std::vector<Idx3>* out) const {
    out->clear();
    for (const auto& kv : block_map)
      if (kv.second->updated & (1u << bit)) out->push_back(kv.first);
  }
  void getAllAllocatedBlocks(std::vector<Idx3>* out) const {
    out->clear();
    out->reserve(block_map.size());
    for (const auto& kv : block_map) out->push_back(kv.first);
  }
  // layer.h:170-182
  void removeDistantBlocks(const Vec3f& center, double max_distance) {
    std::vector<Idx3> erase;
    for (const auto& kv : block_map)
      if (sqnorm(kv.second->origin - center) > max_distance * max_distance)
        erase.push_back(kv.first);
    for (const auto& i : erase) block_map.erase(i);
  }

  float voxel_size;
  size_t voxels_per_side;
  float block_size, voxel_size_inv, block_size_inv, voxels_per_side_inv;
  BlockMap block_map;
};

// ---------------------------------------------------------------------------
// Rigid transform.  minkindr (third party, not under /root/reference; pulled
// unpinned by voxblox_https.rosinstall:16-18) QuatTransformationTemplate<float>
// ::transform = q.rotate(v) + t, and rotate() is Eigen's Quaternion * Vector3:
//   uv = q.vec x v; uv += uv; return v + q.w * uv + q.vec x uv.
// Restated from the published algorithms — parity at this site is unpinned.
// ---------------------------------------------------------------------------
struct Transformation {
  Vec3f t{0, 0, 0};
  float qw = 1, qx = 0, qy = 0, qz = 0;
  Vec3f getPosition() const { return t; }
  Vec3f operator*(const Vec3f& v) const {
    const Vec3f qv{qx, qy, qz};
    Vec3f uv = cross(qv, v);
    uv = uv + uv;
    const Vec3f rot = (v + uv * qw) + cross(qv, uv);
    return rot + t;
  }
};

}  // namespace orc
