// libvbx_hip.so — MI355X (gfx950) TSDF integration hot path of voxblox, hand-written HIP.
//
// What this file implements (reference = /root/reference/voxblox):
//   {Simple,Merged,Fast}TsdfIntegrator::integratePointCloud  src/integrator/tsdf_integrator.cc:242-590
//   RayCaster / ThreadSafeIndex                               src/integrator/integrator_utils.cc
//   Layer<TsdfVoxel> / Block<TsdfVoxel> storage               include/voxblox/core/{layer,block}.h
// behind the C-ABI of include/vbx_hip.h.  See DESIGN.md for the data layout and the kernel list.
//
// Design in one paragraph.  The map lives in HBM as a struct-of-arrays block pool
// (dist[], weight[], rgba[] — vps^3 voxels per block, so a block is three contiguous
// 16 KiB / 16 KiB / 16 KiB runs at vps=16) addressed through an open-addressing hash map
// BlockIndex -> pool slot.  updateTsdfVoxel clamps after every update, so the per-voxel fold
// is neither associative nor commutative: the reference's 1-thread result is
// reproduced by marching every ray once to EMIT (voxel, order) keys, sorting them, and folding
// each voxel's updates sequentially in the reference's point order — parallel across voxels,
// ordered within a voxel.  No MFMA anywhere: this is a raycast/scatter path.
//
// Built with -ffp-contract=off (see vbx_device_math.hpp for why).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <memory_resource>
#include <string>
#include <unordered_map>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "../../include/vbx_hip.h"
#include "vbx_device_math.hpp"

using namespace vbx;

namespace {

// ---------------------------------------------------------------------------
// small host utilities
// ---------------------------------------------------------------------------
thread_local std::string g_create_error;

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      ctx->fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return VBX_ERR_HIP;                                                                  \
    }                                                                                      \
  } while (0)

struct DBuf {  // growable device buffer
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) {
      hipError_t e = hipFree(p);
      if (e != hipSuccess) return e;
      p = nullptr;
      cap = 0;
    }
    size_t want = std::max(bytes, cap + cap / 2);
    want = (want + 255) & ~size_t(255);
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

constexpr uint64_t kEmptyKey = ~0ull;
constexpr uint32_t kInvalidSlot = 0xFFFFFFFFu;

// block flag bits (blk_flags[slot])
constexpr uint32_t kFlagUpdMask = 0x7;      // Update::kMap|kMesh|kEsdf, core/block.h:15-18
constexpr uint32_t kFlagPublished = 0x100;  // block is part of the API-visible Layer
constexpr uint32_t kFlagHasData = 0x200;    // Block::has_data_
constexpr uint32_t kFlagNewThisCall = 0x400;
constexpr uint32_t kFlagEsdfAlloc = 0x1000;   // block exists in Layer<EsdfVoxel>
constexpr uint32_t kFlagEsdfUpdShift = 4;     // ESDF block's Update bits live in bits 4..6
constexpr uint32_t kFlagEsdfPendClassify = 0x2000;  // EsdfIntegrator::updated_blocks_ member (esdf_integrator.cc:54,80)
constexpr uint32_t kFlagEsdfPendOpen = 0x4000;      // holds voxels pushed to open_ by addNewRobotPosition (:84)

// Device-resident scalar state, read back at the per-call sync points.
struct DevState {
  uint32_t pool_used;
  uint32_t free_count;
  uint32_t new_count;
  uint32_t error;  // bit0: pool/hash capacity, bit1: lookup of a missing block
  uint32_t changed;
  uint32_t sentinel_cleared;
  uint32_t blocks_published;
  uint32_t esdf_blocks;
  uint32_t esdf_raise_any;
  uint32_t esdf_relax_blocks;
  uint32_t act_count[3];
  uint32_t fold_long_count;
  uint32_t redo_count;       // rays whose voxel list must be rebuilt after slot assignment
  uint32_t fast_idle_sweep;  // 0xFFFFFFFF - index of the first Fast sweep that found no open ray (0: none yet)
  unsigned long long total_keys;
  unsigned long long voxels_touched;
  unsigned long long rays_cast;
  unsigned long long num_kept;
};

// Host-visible copy of DevState in page-locked, device-mapped host memory.  A read-back is a
// tiny kernel that writes this struct and then its sequence number; the host spins on the
// number.  hipMemcpyAsync + hipStreamSynchronize costs ~35 us per read-back (copy engine launch
// + interrupt-driven wait), and a Fast frame needs five of them.
struct StateMirror {
  DevState st;
  uint32_t extra;
  uint32_t seq;
};

struct MapDev {  // by-value kernel argument
  uint64_t* hkeys;
  uint32_t* hvals;
  uint32_t hmask;
  float* dist;
  float* weight;
  uint32_t* rgba;
  int32_t* blk_idx;     // 3 per slot
  uint32_t* blk_flags;  // 1 per slot
  uint32_t* free_list;
  uint32_t cap_blocks;
  uint32_t nvox;
  int vps;
  int vps_log2;
  float voxel_size;
  float voxel_size_inv;
  float vps_inv;
};

struct CastCfg {  // by-value kernel argument: TsdfIntegratorBase::Config + derived constants
  int exp;  // TEMP experiment switch
  f3 origin;
  float trunc;
  float max_ray_length_m;
  float min_ray_length_m;
  float max_weight;
  float sparsity_factor;
  int carving;
  int allow_clear;
  int use_const_weight;
  int dropoff;
  int sparsity;
  int anti_grazing;
  int max_consecutive;
  float start_factor_times_inv;  // start_voxel_subsampling_factor * voxel_size_inv_
};

struct RayTab {  // SoA ray table indexed by integration order o
  float* px;
  float* py;
  float* pz;      // point_G
  uint32_t* rgba;
  float* w;       // point / bundle weight
  uint8_t* flags; // bit0 cast this ray, bit1 clearing ray
  uint64_t* bkey; // Merged: packed endpoint voxel key of the bundle (anti-grazing), else null
  uint32_t R;
};

__host__ __device__ inline uint64_t pack_block_key(int x, int y, int z) {
  const uint64_t B = 1ull << 20;
  return ((uint64_t)(z + (long long)B) << 42) | ((uint64_t)(y + (long long)B) << 21) |
         (uint64_t)(x + (long long)B);
}
__host__ __device__ inline void unpack_block_key(uint64_t k, int* x, int* y, int* z) {
  const long long B = 1ll << 20;
  *x = (int)((long long)(k & 0x1FFFFF) - B);
  *y = (int)((long long)((k >> 21) & 0x1FFFFF) - B);
  *z = (int)((long long)((k >> 42) & 0x1FFFFF) - B);
}
__host__ __device__ inline uint32_t mix_key(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return (uint32_t)k;
}

// BlockIndex -> pool slot lookup (Layer::getBlockPtrByIndex, layer.h:72-89).
__device__ inline uint32_t map_find(const MapDev& m, uint64_t key) {
  uint32_t h = mix_key(key) & m.hmask;
  for (uint32_t probes = 0; probes <= m.hmask; ++probes) {
    const uint64_t k = __hip_atomic_load(&m.hkeys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == key) return __hip_atomic_load(&m.hvals[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == kEmptyKey) return kInvalidSlot;
    h = (h + 1) & m.hmask;
  }
  return kInvalidSlot;
}

// Insert-if-absent; the pool slot is assigned afterwards by k_assign_slots (temp_block_map_
// + updateLayerWithStoredBlocks, tsdf_integrator.cc:107-126, 137-147).
__device__ inline void map_insert_key(const MapDev& m, uint64_t key, uint32_t* new_list,
                                      DevState* st) {
  uint32_t h = mix_key(key) & m.hmask;
  for (uint32_t probes = 0; probes <= m.hmask; ++probes) {
    uint64_t k = __hip_atomic_load(&m.hkeys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == key) return;
    if (k == kEmptyKey) {
      const unsigned long long old =
          atomicCAS((unsigned long long*)&m.hkeys[h], (unsigned long long)kEmptyKey,
                    (unsigned long long)key);
      if (old == kEmptyKey) {
        const uint32_t i = atomicAdd(&st->new_count, 1u);
        if (i < m.cap_blocks) new_list[i] = h; else atomicOr(&st->error, 1u);
        return;
      }
      if (old == key) return;
    }
    h = (h + 1) & m.hmask;
  }
  atomicOr(&st->error, 1u);
}

// Marks a block as part of the Layer and sets all Update bits (tsdf_integrator.cc:128).  The
// common case — block already published and flagged this frame — is a plain L2 read: only
// the first toucher pays for the atomic, so hundreds of thousands of rays crossing ~200
// blocks do not serialise on ~200 addresses.
__device__ inline void publish_block(const MapDev& m, uint32_t slot, DevState* st) {
  const uint32_t want = kFlagPublished | kFlagUpdMask;
  const uint32_t cur = __hip_atomic_load(&m.blk_flags[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((cur & want) == want) return;
  const uint32_t old = atomicOr(&m.blk_flags[slot], want);
  if (!(old & kFlagPublished)) {
    atomicOr(&m.blk_flags[slot], kFlagNewThisCall);
    atomicAdd(&st->blocks_published, 1u);
  }
}

__device__ inline void unpack_parent_bits(uint32_t s, int* x, int* y, int* z) {
  *x = (int)(int8_t)((s >> 8) & 0xFF);
  *y = (int)(int8_t)((s >> 16) & 0xFF);
  *z = (int)(int8_t)((s >> 24) & 0xFF);
}

// ---------------------------------------------------------------------------
// kernels: map maintenance
// ---------------------------------------------------------------------------
__global__ void k_fill_u64(uint64_t* p, uint64_t v, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

// Per-call counters: one launch instead of several unaligned memsets (each of which the
// runtime splits into head/body/tail fill kernels).
__global__ void k_publish_state(const DevState* st, StateMirror* out, const uint32_t* extra, uint32_t seq) {
  const uint32_t* src = reinterpret_cast<const uint32_t*>(st);
  uint32_t* dst = reinterpret_cast<uint32_t*>(&out->st);
  for (uint32_t i = threadIdx.x; i < sizeof(DevState) / 4; i += blockDim.x) dst[i] = src[i];
  if (threadIdx.x == 0 && extra) out->extra = *extra;
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(&out->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_reset_call_state(DevState* st) {
  st->new_count = 0;
  st->error = 0;
  st->changed = 0;
  st->sentinel_cleared = 0;
  st->blocks_published = 0;
  st->esdf_blocks = 0;
  st->esdf_raise_any = 0;
  st->esdf_relax_blocks = 0;
  st->act_count[0] = st->act_count[1] = st->act_count[2] = 0;
  st->fold_long_count = 0;
  st->fast_idle_sweep = 0;
  st->redo_count = 0;
  st->total_keys = 0;
  st->voxels_touched = 0;
  st->rays_cast = 0;
  st->num_kept = 0;
}

__global__ void k_assign_slots(MapDev m, const uint32_t* new_list, DevState* st) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n = min(st->new_count, m.cap_blocks);
  if (i >= n) return;
  const uint32_t h = new_list[i];
  const uint32_t fc = st->free_count;
  uint32_t slot;
  if (i < fc) {
    slot = m.free_list[fc - 1 - i];
  } else {
    slot = st->pool_used + (i - fc);
  }
  if (slot >= m.cap_blocks) {
    atomicOr(&st->error, 1u);
    return;  // hvals stays invalid; voxels of this block are skipped and the call fails
  }
  int x, y, z;
  unpack_block_key(m.hkeys[h], &x, &y, &z);
  m.blk_idx[3 * slot] = x;
  m.blk_idx[3 * slot + 1] = y;
  m.blk_idx[3 * slot + 2] = z;
  m.blk_flags[slot] = 0;
  __hip_atomic_store(&m.hvals[h], slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void k_commit_alloc(MapDev m, DevState* st) {
  const uint32_t n = min(st->new_count, m.cap_blocks);
  const uint32_t fc = st->free_count;
  if (n <= fc) {
    st->free_count = fc - n;
  } else {
    const uint32_t grow = n - fc;
    st->free_count = 0;
    if (st->pool_used + grow > m.cap_blocks) {
      st->pool_used = m.cap_blocks;
      st->error |= 1u;
    } else {
      st->pool_used += grow;
    }
  }
  st->new_count = 0;
}

// ---------------------------------------------------------------------------
// kernels: ray table construction
// ---------------------------------------------------------------------------
// isPointValid (tsdf_integrator.h:112-129) + T_G_C * point_C + getVoxelWeight
// (tsdf_integrator.cc:231-240); one thread per input point, rows written at the point's
// position in the reference's visiting order (MixedThreadSafeIndex).
__device__ inline bool point_valid(const CastCfg& c, f3 pc, bool freespace, bool* clearing) {
  const float r = f3_norm(pc);
  if (r < c.min_ray_length_m) return false;
  if (r > c.max_ray_length_m) {
    if (c.allow_clear || freespace) {
      *clearing = true;
      return true;
    }
    return false;
  }
  *clearing = freespace;
  return true;
}
__device__ inline float voxel_weight(const CastCfg& c, f3 pc) {
  if (c.use_const_weight) return 1.0f;
  const float dz = fabsf(pc.z);
  if (dz > 1e-6f) return 1.0f / (dz * dz);
  return 0.0f;
}

// SortedThreadSafeIndex (integrator_utils.cc:24-37): visiting order = ascending squared norm.
// key = float bits of point_C.squaredNorm() (non-negative, so they order like the floats) with
// the point index below it: a stable order where the reference's std::sort leaves ties
// unspecified.
__global__ void k_sorted_keys(const float* __restrict__ pts, size_t n, uint64_t* keys) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const f3 pc{pts[3 * p], pts[3 * p + 1], pts[3 * p + 2]};
  keys[p] = ((uint64_t)__float_as_uint(f3_sqnorm(pc)) << 32) | (uint64_t)p;
}
__global__ void k_sorted_inverse(const uint64_t* __restrict__ keys, size_t n, uint32_t* s_of_p) {
  const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  s_of_p[(uint32_t)(keys[s] & 0xFFFFFFFFu)] = (uint32_t)s;
}

__global__ void k_prep_points(const float* __restrict__ pts, const uint32_t* __restrict__ rgba,
                              size_t n, Pose T, CastCfg c, int freespace, RayTab tab,
                              float* pcx, float* pcy, float* pcz, const uint32_t* __restrict__ s_of_p) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const size_t s = s_of_p ? (size_t)s_of_p[p] : mixed_index_inverse(p, n);
  const f3 pc{pts[3 * p], pts[3 * p + 1], pts[3 * p + 2]};
  bool clearing = false;
  const bool valid = point_valid(c, pc, freespace != 0, &clearing);
  const f3 pg = pose_transform(T, pc);
  tab.px[s] = pg.x;
  tab.py[s] = pg.y;
  tab.pz[s] = pg.z;
  tab.rgba[s] = rgba[p];
  tab.w[s] = voxel_weight(c, pc);
  tab.flags[s] = (valid ? 1 : 0) | (clearing ? 2 : 0);
  if (pcx) {  // Merged keeps point_C for the bundle mean
    pcx[s] = pc.x;
    pcy[s] = pc.y;
    pcz[s] = pc.z;
  }
}

// number of rows with the cast flag set (one atomic per workgroup)
__global__ void k_count_cast(const uint8_t* __restrict__ flags, uint32_t n, DevState* st) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = __syncthreads_count(i < n && (flags[i] & 1));
  if (threadIdx.x == 0 && c) atomicAdd(&st->rays_cast, (unsigned long long)c);
}

// ---------------------------------------------------------------------------
// kernels: generic ray march over a ray table
// ---------------------------------------------------------------------------
__device__ inline bool ray_init(RayCaster& rc, const RayTab& tab, uint32_t o, const CastCfg& c,
                                const MapDev& m, bool from_origin, f3* pg_out) {
  const uint8_t fl = tab.flags[o];
  if (!(fl & 1)) return false;
  const f3 pg{tab.px[o], tab.py[o], tab.pz[o]};
  rc.init(c.origin, pg, (fl & 2) != 0, c.carving != 0, c.max_ray_length_m, m.voxel_size_inv,
          c.trunc, from_origin);
  if (pg_out) *pg_out = pg;
  return true;
}

// cnt[o] = number of voxel indices the ray emits (ray_length_in_steps_ + 1), or `limit[o]`.
__global__ void k_ray_count(RayTab tab, CastCfg c, MapDev m, int from_origin,
                            const uint32_t* __restrict__ limit, uint32_t* cnt) {
  const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o > tab.R) return;
  if (o == tab.R) {
    cnt[o] = 0;
    return;
  }
  RayCaster rc;
  uint32_t n = 0;
  if (ray_init(rc, tab, o, c, m, from_origin != 0, nullptr)) {
    n = (rc.cur == 0) ? rc.steps + 1 : 0;
    if (limit) n = min(n, limit[o]);
  }
  cnt[o] = n;
}

// Walks every ray and makes sure each block it crosses has a pool slot
// (allocateStorageAndGetVoxelPtr's block part, tsdf_integrator.cc:97-126).
__global__ void k_ray_mark_blocks(RayTab tab, CastCfg c, MapDev m, int from_origin,
                                  const uint32_t* __restrict__ limit, uint32_t* new_list,
                                  DevState* st) {
  const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= tab.R) return;
  RayCaster rc;
  if (!ray_init(rc, tab, o, c, m, from_origin != 0, nullptr)) return;
  if (rc.cur != 0) return;
  const uint32_t n = min(limit ? limit[o] : 0xFFFFFFFFu, rc.steps + 1);
  BlockWalk bw;
  bw.start(rc, m.vps, m.vps_inv);
  for (uint32_t k = 0; k < n; ++k) {
    if (bw.entered) map_insert_key(m, pack_block_key(bw.bx, bw.by, bw.bz), new_list, st);
    bw.step(m.vps, m.vps_log2);
  }
}

// The ray march proper: every visited voxel becomes one 64-bit key
//   (pool_slot * nvox + linear_index) << 32 | order
// written at off[o] + k.  Blocks touched are published and get all Update bits
// (tsdf_integrator.cc:128).  Merged's anti-grazing test (:415-422) is a binary search in
// the sorted bundle keys.  The walk is BlockWalk (branch-free steps, no int64 index math).
__global__ void k_ray_emit(RayTab tab, CastCfg c, MapDev m, int from_origin,
                           const uint32_t* __restrict__ limit, const uint32_t* __restrict__ off,
                           uint64_t* keys, const uint64_t* __restrict__ graze_keys,
                           uint32_t n_graze, DevState* st) {
  const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= tab.R) return;
  RayCaster rc;
  if (!ray_init(rc, tab, o, c, m, from_origin != 0, nullptr)) return;
  if (rc.cur != 0) return;
  const uint32_t n = min(limit ? limit[o] : 0xFFFFFFFFu, rc.steps + 1);
  const bool clearing = (tab.flags[o] & 2) != 0;
  BlockWalk bw;
  bw.start(rc, m.vps, m.vps_inv);
  bool need_lookup = false;
  uint32_t slot = kInvalidSlot;
  const uint32_t base = off[o];
  const uint32_t lmask = (uint32_t)m.vps - 1u;
  for (uint32_t k = 0; k < n; ++k) {
    uint64_t out = ~0ull;  // sorts last, skipped by the fold
    need_lookup = need_lookup || bw.entered;
    bool skip = false;
    if (graze_keys) {
      // voxel_map.find(global_voxel_idx) != end && (clearing || idx != kv.first)
      const long long gx = (long long)bw.bx * m.vps + (long long)(bw.lin & lmask);
      const long long gy = (long long)bw.by * m.vps + (long long)((bw.lin >> m.vps_log2) & lmask);
      const long long gz = (long long)bw.bz * m.vps + (long long)((bw.lin >> (2 * m.vps_log2)) & lmask);
      const uint64_t vk = ((uint64_t)(gz + (1ll << 20)) << 42) | ((uint64_t)(gy + (1ll << 20)) << 21) |
                          (uint64_t)(gx + (1ll << 20));
      if (clearing || vk != tab.bkey[o]) {
        uint32_t lo = 0, hi = n_graze;
        while (lo < hi) {
          const uint32_t mid = (lo + hi) >> 1;
          if (graze_keys[mid] < vk) lo = mid + 1; else hi = mid;
        }
        skip = (lo < n_graze && graze_keys[lo] == vk);
      }
    }
    if (!skip) {
      if (need_lookup) {
        need_lookup = false;
        slot = map_find(m, pack_block_key(bw.bx, bw.by, bw.bz));
        if (slot == kInvalidSlot) {
          atomicOr(&st->error, 2u);
        } else {
          publish_block(m, slot, st);
        }
      }
      if (slot != kInvalidSlot) out = ((uint64_t)(slot * m.nvox + bw.lin) << 32) | o;
    }
    keys[base + k] = out;
    bw.step(m.vps, m.vps_log2);
  }
}

// ---------------------------------------------------------------------------
// kernel: ordered per-voxel fold == updateTsdfVoxel (tsdf_integrator.cc:150-209) applied to
// each voxel's updates in ascending integration order.  One thread per segment head.
// ---------------------------------------------------------------------------
__device__ inline void tsdf_update(const CastCfg& c, float voxel_size, f3 pg, l3 g,
                                   uint32_t color, float weight, float& d, float& W,
                                   uint32_t& col) {
  const f3 center = center_point_from_grid_index(g, voxel_size);
  // computeDistance, tsdf_integrator.cc:216-228
  const f3 a = f3_sub(center, c.origin);
  const f3 b = f3_sub(pg, c.origin);
  const float dist_G = f3_norm(b);
  const float dist_G_V = f3_dot(a, b) / dist_G;
  const float sdf = dist_G - dist_G_V;

  float uw = weight;
  const float eps = voxel_size;
  if (c.dropoff && sdf < -eps) {
    uw = weight * (c.trunc + sdf) / (c.trunc - eps);
    uw = std_max(uw, 0.0f);
  }
  if (c.sparsity) {
    if (fabsf(sdf) < c.trunc) uw *= c.sparsity_factor;
  }
  const float nw = W + uw;
  if (nw < 1e-6f) return;
  const float nsdf = (sdf * uw + d * W) / nw;
  if (fabsf(sdf) < c.trunc) col = blend_two_colors(col, W, color, uw);
  d = (nsdf > 0.0f) ? std_min(c.trunc, nsdf) : std_max(-c.trunc, nsdf);
  W = std_min(c.max_weight, nw);
}

// Everything of updateTsdfVoxel that does not depend on the voxel's state (tsdf_integrator.cc:
// 157-183): the projective sdf and the (drop-off / sparsity adjusted) weight of one update.
__device__ inline void tsdf_update_inputs(const CastCfg& c, float voxel_size, f3 pg, l3 g, float weight,
                                          float* sdf_out, float* uw_out) {
  const f3 center = center_point_from_grid_index(g, voxel_size);
  const f3 a = f3_sub(center, c.origin);
  const f3 b = f3_sub(pg, c.origin);
  const float dist_G = f3_norm(b);
  const float dist_G_V = f3_dot(a, b) / dist_G;
  const float sdf = dist_G - dist_G_V;
  float uw = weight;
  const float eps = voxel_size;
  if (c.dropoff && sdf < -eps) {
    uw = weight * (c.trunc + sdf) / (c.trunc - eps);
    uw = std_max(uw, 0.0f);
  }
  if (c.sparsity) {
    if (fabsf(sdf) < c.trunc) uw *= c.sparsity_factor;
  }
  *sdf_out = sdf;
  *uw_out = uw;
}
// The state-dependent rest (tsdf_integrator.cc:188-208).
__device__ inline void tsdf_update_state(const CastCfg& c, float sdf, float uw, uint32_t color, float& d,
                                         float& W, uint32_t& col) {
  const float nw = W + uw;
  if (nw < 1e-6f) return;
  const float nsdf = (sdf * uw + d * W) / nw;
  if (fabsf(sdf) < c.trunc) col = blend_two_colors(col, W, color, uw);
  d = (nsdf > 0.0f) ? std_min(c.trunc, nsdf) : std_max(-c.trunc, nsdf);
  W = std_min(c.max_weight, nw);
}

constexpr uint32_t kFoldShort = 48;  // longer runs go to the wave-cooperative kernel

__device__ inline l3 voxel_of_gid(const MapDev& m, uint32_t gid) {
  const uint32_t slot = gid / m.nvox;
  const uint32_t lin = gid - slot * m.nvox;
  const int lx = lin & (m.vps - 1);
  const int ly = (lin >> m.vps_log2) & (m.vps - 1);
  const int lz = lin >> (2 * m.vps_log2);
  return {(long long)m.blk_idx[3 * slot] * m.vps + lx, (long long)m.blk_idx[3 * slot + 1] * m.vps + ly,
          (long long)m.blk_idx[3 * slot + 2] * m.vps + lz};
}

__global__ void k_fold(const uint64_t* __restrict__ keys, size_t n, RayTab tab, CastCfg c,
                       MapDev m, uint32_t* long_list, DevState* st) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t key = (i < n) ? keys[i] : ~0ull;
  const uint32_t gid = (uint32_t)(key >> 32);
  const bool head = (key != ~0ull) && !(i > 0 && (uint32_t)(keys[i - 1] >> 32) == gid);
  const int nheads = __syncthreads_count(head);
  if (threadIdx.x == 0 && nheads) atomicAdd(&st->voxels_touched, (unsigned long long)nheads);
  if (!head) return;  // only segment heads fold

  // a long run (the keys of a voxel are contiguous, so one look ahead tells): hand it to
  // k_fold_long untouched
  if (i + kFoldShort < n && (uint32_t)(keys[i + kFoldShort] >> 32) == gid) {
    const uint32_t o = atomicAdd(&st->fold_long_count, 1u);
    long_list[o] = (uint32_t)i;
    return;
  }
  const l3 g = voxel_of_gid(m, gid);
  float d = m.dist[gid];
  float W = m.weight[gid];
  uint32_t col = m.rgba[gid];
  size_t j = i;
  uint64_t kj = key;
  while (true) {
    const uint32_t o = (uint32_t)(kj & 0xFFFFFFFFu);
    const f3 pg{tab.px[o], tab.py[o], tab.pz[o]};
    tsdf_update(c, m.voxel_size, pg, g, tab.rgba[o], tab.w[o], d, W, col);
    ++j;
    if (j >= n) break;
    kj = keys[j];
    if ((uint32_t)(kj >> 32) != gid) break;
  }
  m.dist[gid] = d;
  m.weight[gid] = W;
  m.rgba[gid] = col;
}

// Long runs (the voxels around the sensor origin collect one update per ray): one wave per run,
// 64 updates per step.  The state-independent part of the 64 updates (sdf, weight) is computed
// in parallel; the ordered fold over them is then done by the cheapest exact method:
//   1. every update is a no-op on the current state (saturated free-space voxel)  -> skip;
//   2. the distance provably stays where it is (clamped at +-trunc) and no colour changes:
//      only the weight chain W <- min(max_weight, W + w) is evaluated in order, then all 64
//      distance updates are verified in parallel against their own W;
//   3. otherwise the 64 updates are applied in order (operands broadcast lane by lane).
// All three produce exactly the sequential result of updateTsdfVoxel.
__global__ void __launch_bounds__(256) k_fold_long(const uint64_t* __restrict__ keys, size_t n, RayTab tab,
                                                   CastCfg c, MapDev m, const uint32_t* __restrict__ long_list,
                                                   DevState* st) {
  const int lane = threadIdx.x & 63;
  const uint32_t n_long = st->fold_long_count;
  const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t seg = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; seg < n_long; seg += n_waves) {
    const size_t i0 = long_list[seg];
    const uint32_t gid = (uint32_t)(keys[i0] >> 32);
    const l3 g = voxel_of_gid(m, gid);
    float d = m.dist[gid];
    float W = m.weight[gid];
    uint32_t col = m.rgba[gid];
    // kU chunks of 64 updates are fetched together (the gathers of px/py/pz/w/rgba by ray index
    // cost ~2 us of latency per chunk when issued one chunk at a time, and a run of 300k updates
    // on the sensor's own voxel is 4800 chunks), then folded chunk by chunk in order.
    constexpr int kU = 4;
    bool more = true;
    for (size_t base = i0; more; base += 64 * kU) {
      float sdf_u[kU], uw_u[kU];
      uint32_t color_u[kU];
      int cnt_u[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const size_t i = base + 64 * u + lane;
        const uint64_t key = (i < n) ? keys[i] : ~0ull;
        const bool mine = (key != ~0ull) && ((uint32_t)(key >> 32) == gid);
        const unsigned long long V = __ballot(mine);
        // keys of one voxel are contiguous: the valid lanes are a prefix
        cnt_u[u] = (V == ~0ull) ? 64 : (__ffsll((long long)~V) - 1);
        sdf_u[u] = 0.f; uw_u[u] = 0.f; color_u[u] = 0;
        if (lane < cnt_u[u]) {
          const uint32_t o = (uint32_t)(key & 0xFFFFFFFFu);
          const f3 pg{tab.px[o], tab.py[o], tab.pz[o]};
          tsdf_update_inputs(c, m.voxel_size, pg, g, tab.w[o], &sdf_u[u], &uw_u[u]);
          color_u[u] = tab.rgba[o];
        }
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int cnt = cnt_u[u];
        if (cnt == 0) { more = false; break; }
        const float sdf = sdf_u[u], uw = uw_u[u];
        const uint32_t color = color_u[u];
        const bool inband = fabsf(sdf) < c.trunc;
        // 1. identity test against the current state
        bool same = true;
        if (lane < cnt) {
          float d1 = d, W1 = W;
          uint32_t c1 = col;
          tsdf_update_state(c, sdf, uw, color, d1, W1, c1);
          same = (__float_as_uint(d1) == __float_as_uint(d)) && (__float_as_uint(W1) == __float_as_uint(W)) && (c1 == col);
        }
        if (!__all(same)) {
          // 2. weight chain + parallel verification that d does not move
          bool done = false;
          if (!__any(lane < cnt && inband)) {
            // the chain itself: operands come out of the lanes with v_readlane (constant lane
            // index, fully unrolled) — a ds_bpermute per element made this loop the whole cost
            // of the fold (~2.7 us per 64 updates)
            float Wrun = W, Wmine = W;
            if (cnt == 64 && W >= 1e-6f && __all(uw >= 0.0f)) {
              // full chunk, weights only grow: the `new_weight < kFloatEpsilon` exit of
              // updateTsdfVoxel (tsdf_integrator.cc:183-186) cannot trigger, so the chain is two
              // dependent VALU ops per update with no branches
#pragma unroll
              for (int j = 0; j < 64; ++j) {
                const float uwj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(uw), j));
                Wmine = (lane == j) ? Wrun : Wmine;
                Wrun = std_min(c.max_weight, Wrun + uwj);
              }
            } else {
#pragma unroll
              for (int j = 0; j < 64; ++j) {
                if (j < cnt) {
                  const float uwj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(uw), j));
                  if (lane == j) Wmine = Wrun;
                  const float nw = Wrun + uwj;
                  if (!(nw < 1e-6f)) Wrun = std_min(c.max_weight, nw);
                }
              }
            }
            bool ok = true;
            if (lane < cnt) {
              float d1 = d, W1 = Wmine;
              uint32_t c1 = col;
              tsdf_update_state(c, sdf, uw, color, d1, W1, c1);
              ok = (__float_as_uint(d1) == __float_as_uint(d));
            }
            if (__all(ok)) {
              W = Wrun;
              done = true;
            }
          }
          // 3. generic ordered application
          if (!done) {
#pragma unroll
            for (int j = 0; j < 64; ++j) {
              if (j < cnt) {
                const float sj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sdf), j));
                const float wj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(uw), j));
                const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)color, j);
                tsdf_update_state(c, sj, wj, cj, d, W, col);
              }
            }
          }
        }
        if (cnt < 64) { more = false; break; }
      }
    }
    if (lane == 0) {
      m.dist[gid] = d;
      m.weight[gid] = W;
      m.rgba[gid] = col;
    }
  }
}

// ---------------------------------------------------------------------------
// kernels: Merged integrator bundling (tsdf_integrator.cc:340-407)
// ---------------------------------------------------------------------------
// key[s] = clearing << 63 | packed endpoint voxel index; invalid points sort last.
__global__ void k_merged_keys(RayTab pt, uint32_t n, MapDev m, uint64_t* keys, uint32_t* vals) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const uint8_t fl = pt.flags[s];
  uint64_t key = ~0ull;
  if (fl & 1) {
    const l3 g = grid_index_from_point({pt.px[s], pt.py[s], pt.pz[s]}, m.voxel_size_inv);
    key = ((uint64_t)(g.z + (1ll << 20)) << 42) | ((uint64_t)(g.y + (1ll << 20)) << 21) |
          (uint64_t)(g.x + (1ll << 20));
    if (fl & 2) key |= 1ull << 63;
  }
  keys[s] = key;
  vals[s] = s;
}

// head[i] = 1 where a new bundle starts in the sorted (key, s) list.
__global__ void k_merged_heads(const uint64_t* __restrict__ keys, uint32_t n, uint32_t* head) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  if (i == n) {
    head[i] = 0;
    return;
  }
  const uint64_t k = keys[i];
  head[i] = (k != ~0ull && (i == 0 || keys[i - 1] != k)) ? 1u : 0u;
}

// One thread per bundle: running weighted mean of point_C, blended colour, summed weight in
// push_back (= visiting) order; clearing bundles take their first usable point only
// (tsdf_integrator.cc:387-405); merged_point_G = T_G_C * merged_point_C (:407).
__global__ void k_merged_bundle(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                const uint32_t* __restrict__ head, const uint32_t* __restrict__ rank,
                                uint32_t n, RayTab pt, const float* __restrict__ pcx,
                                const float* __restrict__ pcy, const float* __restrict__ pcz,
                                Pose T, RayTab out, uint64_t* graze_keys, const uint32_t* __restrict__ perm,
                                DevState* st) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !head[i]) return;
  const uint64_t key = keys[i];
  const bool clearing = (key >> 63) != 0;
  const uint32_t br = rank[i];                    // rank in ascending key order
  const uint32_t b = perm ? perm[br] : br;        // row = position in the visiting order of the bundles
  f3 mp{0.f, 0.f, 0.f};
  uint32_t mc = 0;
  float mw = 0.0f;
  for (uint32_t j = i; j < n && keys[j] == key; ++j) {
    const uint32_t s = vals[j];
    const float pw = pt.w[s];
    if (pw < 1e-6f) continue;
    const f3 pc{pcx[s], pcy[s], pcz[s]};
    const float tw = mw + pw;
    mp = {(mp.x * mw + pc.x * pw) / tw, (mp.y * mw + pc.y * pw) / tw, (mp.z * mw + pc.z * pw) / tw};
    mc = blend_two_colors(mc, mw, pt.rgba[s], pw);
    mw += pw;
    if (clearing) break;
  }
  const f3 pg = pose_transform(T, mp);
  out.px[b] = pg.x;
  out.py[b] = pg.y;
  out.pz[b] = pg.z;
  out.rgba[b] = mc;
  out.w[b] = mw;
  out.flags[b] = 1 | (clearing ? 2 : 0);
  out.bkey[b] = key & ~(1ull << 63);
  if (!clearing && graze_keys) graze_keys[br] = key;  // stays sorted: binary-searched by the march
  (void)st;
}

// Per bundle (ascending key rank): its key and the visiting position of its first point — the
// order in which bundleRays inserts the keys into its unordered_map.
__global__ void k_merged_collect(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                 const uint32_t* __restrict__ head, const uint32_t* __restrict__ rank, uint32_t n,
                                 uint64_t* bkeys, uint32_t* first_s) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !head[i]) return;
  bkeys[rank[i]] = keys[i];
  first_s[rank[i]] = vals[i];
}

// ---------------------------------------------------------------------------
// kernels: Fast integrator (tsdf_integrator.cc:488-590)
// ---------------------------------------------------------------------------
// start_voxel_approx_set_.replaceHash(cell at start_voxel_subsampling_factor x resolution),
// tsdf_integrator.cc:514-519.  key = slot << 32 | s so that a stable radix sort groups the
// probes of one ApproxHashSet slot in visiting order; val = the 32-bit hash.
__global__ void k_fast_keys(RayTab pt, uint32_t n, CastCfg c, uint64_t* keys, uint32_t* vals) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  uint64_t key = ~0ull;
  uint32_t h = 0;
  if (pt.flags[s] & 1) {
    const l3 g = grid_index_from_point({pt.px[s], pt.py[s], pt.pz[s]}, c.start_factor_times_inv);
    h = long_index_hash(g);
    key = ((uint64_t)(h & 0xFFFFFu) << 32) | s;
  }
  keys[s] = key;
  vals[s] = h;
}

// Exact replay of ApproxHashSet<20,10000>::replaceHash over the sorted probes
// (approx_hash_array.h:125-134): a probe "replaces" iff the value it finds in its slot —
// the previous probe's hash, or the slot's content from before this frame — differs from
// its own hash; every probe leaves its hash behind.  set_vals mirrors pseudo_set_ at
// offset_ (u32 per slot; the u64 max() sentinel of slot 0 is tracked separately).
__global__ void k_fast_start_dedupe(const uint64_t* __restrict__ keys,
                                    const uint32_t* __restrict__ vals, uint32_t n,
                                    const uint32_t* __restrict__ set_vals, uint32_t offset,
                                    int sentinel_live, uint8_t* flags_by_s) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t key = keys[i];
  if (key == ~0ull) return;
  const uint32_t slot = (uint32_t)(key >> 32);
  const uint32_t s = (uint32_t)(key & 0xFFFFFFFFu);
  const uint32_t h = vals[i];
  bool replaced;
  if (i > 0 && (uint32_t)(keys[i - 1] >> 32) == slot) {
    replaced = (vals[i - 1] != h);
  } else {
    const uint32_t ai = slot + offset;
    if (ai == 0 && sentinel_live) replaced = true;  // slot holds size_t max()
    else replaced = (set_vals[ai] != h);
  }
  if (!replaced) flags_by_s[s] &= ~1;  // `continue` at tsdf_integrator.cc:517-519
}
// Second half of the replay: the last probe of every slot leaves its hash in the set
// (separate launch so that no thread can read a slot after this frame has written it).
__global__ void k_fast_start_commit(const uint64_t* __restrict__ keys,
                                    const uint32_t* __restrict__ vals, uint32_t n,
                                    uint32_t* set_vals, uint32_t offset, DevState* st) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t key = keys[i];
  if (key == ~0ull) return;
  const uint32_t slot = (uint32_t)(key >> 32);
  const bool last = (i + 1 >= n) || ((uint32_t)(keys[i + 1] >> 32) != slot);
  if (last) {
    set_vals[slot + offset] = vals[i];
    if (slot + offset == 0) st->sentinel_cleared = 1;
  }
}

// Compacts the rays that survive the start-voxel test, keeping visiting order.
__global__ void k_compact_flags(const uint8_t* __restrict__ flags, uint32_t n, uint32_t* keep) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > n) return;
  keep[s] = (s < n && (flags[s] & 1)) ? 1u : 0u;
}
__global__ void k_compact_rays(RayTab in, const uint32_t* __restrict__ keep,
                               const uint32_t* __restrict__ pos, uint32_t n, RayTab out,
                               DevState* st) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  if (s == n - 1) st->num_kept = pos[s] + keep[s];
  if (!keep[s]) return;
  const uint32_t r = pos[s];
  out.px[r] = in.px[s];
  out.py[r] = in.py[s];
  out.pz[r] = in.pz[s];
  out.rgba[r] = in.rgba[s];
  out.w[r] = in.w[s];
  out.flags[r] = in.flags[s];
}

// Per-ray voxel lists: vox[off[r] + k] = pool_slot * nvox + linear_index of the k-th voxel the
// ray visits walking from the surface towards the sensor (cast_from_origin = false,
// tsdf_integrator.cc:521-525).  Built once per frame; the solver and the emit step then work
// on these lists instead of re-running the DDA and the block hash lookups.
constexpr int kListRPW = 64;  // rays per wave in k_fast_build_lists
template <int RPW>
__global__ void __launch_bounds__(256)
k_fast_build_lists(RayTab tab, CastCfg c, MapDev m, const uint32_t* __restrict__ off,
                   uint32_t* vox, uint32_t vox_cap, uint32_t* new_list, const uint32_t* __restrict__ redo_in,
                   uint32_t* redo_out, DevState* st) {
  // First pass (redo_in == nullptr): blocks met for the first time are inserted into the map
  // here (the block part of allocateStorageAndGetVoxelPtr, tsdf_integrator.cc:97-126); they
  // only get their pool slot after this kernel, so a ray that crossed one is queued in
  // redo_out and rebuilt by the second pass (redo_in = that queue).  In steady state a frame
  // adds a handful of blocks, so the second pass touches a few hundred rays.
  // RPW rays per wave, one per lane in the low lanes: the walk is a serial dependency chain per
  // ray, so with all 64 lanes busy the 70k rays of a frame are ~1 wave per SIMD and nothing
  // hides the latencies; fewer rays per wave means more resident waves.  Every ray lane stages
  // 16 list entries in LDS, then ALL 64 lanes write them out ray by ray as 64-byte runs (4 rays
  // per store instruction) instead of scattered dwords.
  __shared__ uint32_t s_buf[4][RPW][17];  // [wave][ray][entry], padded against bank conflicts
  __shared__ uint64_t s_keys[4][RPW][17]; // keys, then pool slots, of the blocks a ray enters within a chunk
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const uint32_t limit = redo_in ? min(st->redo_count, tab.R) : tab.R;
  const uint32_t stride = gridDim.x * (blockDim.x / 64) * RPW;
  // grid-stride over the work items; the trip count is uniform within a wave (the flush below
  // is wave-cooperative)
  for (uint32_t wbase = (blockIdx.x * (blockDim.x / 64) + wv) * RPW; wbase < limit; wbase += stride) {
    const uint32_t tix = wbase + lane;
    const bool in_range = lane < RPW && tix < limit;
    const uint32_t r = in_range ? (redo_in ? redo_in[tix] : tix) : 0;
    RayCaster rc;
    bool live = in_range && ray_init(rc, tab, r, c, m, /*from_origin=*/false, nullptr) && rc.cur == 0;
    const uint32_t base = in_range ? off[r] : 0;
    uint32_t len = live ? rc.steps + 1 : 0;
    if (live && base + len > vox_cap) {  // cannot happen while the host's per-ray bound holds
      atomicOr(&st->error, 4u);
      len = 0;
    }
    BlockWalk bw{};
    if (live) bw.start(rc, m.vps, m.vps_inv);
    uint32_t slot = kInvalidSlot;
    bool redo = false;
    auto lookup = [&](uint64_t key) -> uint32_t {
      const uint32_t sl = map_find(m, key);
      if (sl == kInvalidSlot) {
        if (redo_in) {
          atomicOr(&st->error, 2u);
        } else {
          map_insert_key(m, key, new_list, st);
          redo = true;
        }
      }
      return sl;
    };
    for (uint32_t k0 = 0; __any(k0 < len); k0 += 16) {
      // (a) 16 DDA steps, no memory traffic: linear voxel index + "enters a new block" mark
      // per entry, the new blocks' keys on the side.  A hash lookup inside this loop would
      // stall the whole wave at almost every step (some lane always crosses a block face).
      uint64_t* tkeys = s_keys[wv][lane < RPW ? lane : 0];  // at most one block change per step
      int nt = 0;
      for (int j = 0; j < 16; ++j) {
        uint32_t e = 0xFFFFFFFFu;
        if (k0 + j < len) {
          e = bw.lin;
          if (bw.entered) {
            tkeys[nt] = pack_block_key(bw.bx, bw.by, bw.bz);
            e |= 0x80000000u | ((uint32_t)nt << 24);
            ++nt;
          }
          bw.step(m.vps, m.vps_log2);
        }
        if (lane < RPW) s_buf[wv][lane][j] = e;
      }
      // (b) the lookups, rank by rank: all lanes issue their t-th lookup together
      for (int t = 0; __any(t < nt); ++t)
        if (t < nt) tkeys[t] = (c.exp & 1) ? 0ull : (uint64_t)lookup(tkeys[t]);
      // (c) entries -> global voxel ids
      for (int j = 0; j < 16; ++j) {
        const uint32_t e = (lane < RPW) ? s_buf[wv][lane][j] : 0xFFFFFFFFu;
        uint32_t gid = 0xFFFFFFFFu;
        if (e != 0xFFFFFFFFu) {
          if (e & 0x80000000u) {
            slot = (uint32_t)tkeys[(e >> 24) & 0x7Fu];
          }
          if (slot != kInvalidSlot) gid = slot * m.nvox + (e & 0xFFFFFFu);
        }
        if (lane < RPW) s_buf[wv][lane][j] = gid;
      }
      // wave-synchronous flush (same wave wrote and reads; LDS ops of one wave are ordered)
      const int sub = lane >> 4, e = lane & 15;
      for (int q = 0; q < RPW / 4; ++q) {
        const int src = q * 4 + sub;  // lane whose ray is being written
        const uint32_t sbase = __shfl(base, src);
        const uint32_t slen = __shfl(len, src);
        if (k0 + e < slen && !(c.exp & 2)) vox[sbase + k0 + e] = s_buf[wv][src][e];
      }
    }
    if (redo) redo_out[atomicAdd(&st->redo_count, 1u)] = r;
  }
}

// Early-termination solver.  With an exact observed-set, "voxel already observed when ray r
// probes it" == "some ray r' < r reaches that voxel", i.e. owner(v) = min{r' reaching v} < r.
// Which voxels a ray reaches depends on where it terminates, which depends on the owners of
// the voxels ahead of it — a fixed point, unique because dependencies only run from lower to
// higher r.  One sweep = every ray re-evaluates its termination against the owners of the
// previous sweep and publishes the owners for the next one (atomicMin); repeat until no
// termination step moves.  Owner entries carry a descending sweep tag in their high bits so
// the two ping-pong arrays never need clearing.  (tsdf_integrator.cc:531-551)
//
// One wave per ray, 64 probes per step: the lanes fetch 64 consecutive owners of the ray's
// voxel list at once, the consecutive-collision counter becomes a run-length computed from
// the ballot mask, and the first lane whose run exceeds max_consecutive_ray_collisions is
// the termination step.
// Two-sided form of that iteration.  Every ray carries a lower bound TL and an upper bound TH
// on its true number of probes T* (TL = 0, TH = full path length to start with).
//   certain claims  CL(v) = min{r : v among the first TL_r voxels of r}   (>= true owner)
//   possible claims CH(v) = min{r : v among the first TH_r voxels of r}   (<= true owner)
// A sweep recomputes TH from the certain claims only (fewest collisions -> latest stop) and TL
// from the possible claims (most collisions -> earliest stop).  TL only grows and TH only
// shrinks, so CL is a persistent atomicMin array, a ray with TL == TH is final for good and
// drops out of the work list, and only the possible claims of the still-open rays are rebuilt
// per sweep (tagged ping-pong arrays; the final rays' claims are already in CL).  The loop
// ends when no ray is open; the fixed point is the reference's sequential result.
//
// G lanes per ray: the lanes fetch G consecutive entries of the ray's voxel list at once, the
// consecutive-collision counter becomes a run-length computed from the ballot mask, and the
// first lane whose run exceeds max_consecutive_ray_collisions is the termination step.
// atomicMin that first looks: near the sensor origin tens of thousands of rays share the same
// few voxels, and same-address atomics serialise (~90 per microsecond on one word).  Claims
// only ever decrease, so when the word already holds a smaller value the RMW is a no-op and
// can be skipped after an L2 read.
__device__ inline void claim_min(uint32_t* p, uint32_t val) {
  if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > val) atomicMin(p, val);
}

struct SweepArgs {
  const uint32_t* off;      // voxel list offsets (R+1)
  const uint32_t* vox;      // voxel lists
  const uint32_t* list_in;  // open rays of this sweep (null: identity, all R rays)
  uint32_t* list_out;       // open rays for the next sweep
  uint32_t n_in;            // upper bound of the input list length (grid size)
  int cnt_in, cnt_out;      // DevState::act_count[] indices of the input / output list lengths
  uint32_t* cl;             // certain claims (persistent within the frame)
  const uint32_t* ch_rd;    // possible claims of open rays, previous sweep
  uint32_t* ch_wr;          // possible claims of open rays, this sweep
  uint32_t tag_cl, tag_rd, tag_wr;
  int s_bits;
  int max_consecutive;
  uint32_t* TL; uint32_t* TH; uint32_t* U;
  const uint32_t* obs;      // voxels observed in earlier frames since the last reset (or null)
  uint32_t obs_epoch;
  uint32_t sweep_idx;       // 0, 1, 2, ... within the frame
  int init;                 // 1: first pass (publish full-path possible claims, no reads)
  int l_only;               // 1: only tighten the lower bounds (TH and the possible claims stay)
};

// One sweep step for the ray handled by this lane group.  All 64 lanes of the wave must call
// it together.  Returns (on the group's lane 0) whether the ray is still open.
template <int G, bool kCoherentReads>
__device__ inline bool sweep_ray(const SweepArgs& a, bool ray_ok, uint32_t r, int grp, int gl) {
  const uint32_t beg = ray_ok ? a.off[r] : 0;
  const uint32_t len = ray_ok ? a.off[r + 1] - beg : 0;
  const uint32_t smask = (1u << a.s_bits) - 1;
  const uint32_t cl_val = (a.tag_cl << a.s_bits) | r;
  const uint32_t ch_val = (a.tag_wr << a.s_bits) | r;
  const unsigned long long gmask = (G == 64) ? ~0ull : ((1ull << G) - 1ull);
  const unsigned long long below = (gl == 63) ? ~0ull : ((2ull << gl) - 1ull);
  const uint32_t tl_old = (ray_ok && !a.init) ? a.TL[r] : 0;
  int consL = 0, consH = 0;  // carries of the two collision counters
  uint32_t tl = len, th = len;
  bool brokeL = false, brokeH = false;
  bool doneL = (len == 0) || a.init, done = (len == 0);
  if (a.init) tl = 0;
  if (a.l_only) th = ray_ok ? a.TH[r] : 0;
  // No bound can move below the old lower bound: under the (shrinking) possible claims the ray
  // did not stop before probe kT = tl_old - 1, so neither does it under the certain ones, and a
  // collision run that ends at kT or later starts at kT - max_consecutive at the earliest.
  // The scan therefore restarts there with clear counters; the claims of the skipped prefix
  // are already in CL.
  const uint32_t k0 = (tl_old > (uint32_t)a.max_consecutive + 1u) ? tl_old - 1u - (uint32_t)a.max_consecutive : 0u;
  // The list entries of the next step are fetched while the current step's claim reads are in
  // flight, and both claim words are read unconditionally: one memory latency per step instead
  // of three dependent ones (vox -> cl -> ch).
  uint32_t gid_pf = (k0 + gl < len) ? a.vox[beg + k0 + gl] : 0xFFFFFFFFu;
  for (uint32_t base = k0; __any(!done); base += G) {
    const uint32_t k = base + gl;
    const bool act = !done && k < len;
    const uint32_t gid = act ? gid_pf : 0xFFFFFFFFu;
    gid_pf = (!done && k + G < len) ? a.vox[beg + k + G] : 0xFFFFFFFFu;
    bool pL = false, pH = false;  // collision under certain / possible claims
    if (gid != 0xFFFFFFFFu && !a.init) {
      const uint32_t c1 = __hip_atomic_load(&a.cl[gid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t c2 = kCoherentReads
                              ? __hip_atomic_load(&a.ch_rd[gid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                              : a.ch_rd[gid];
      const uint32_t c3 = a.obs ? a.obs[gid] : 0u;
      pL = ((c1 >> a.s_bits) == a.tag_cl) && ((c1 & smask) < r);
      if (a.obs) pL = pL || (c3 == a.obs_epoch);  // seen in an earlier frame of this epoch
      pH = pL || (((c2 >> a.s_bits) == a.tag_rd) && ((c2 & smask) < r));
    }
    // upper bound TH: stop on a run of certain collisions
    const unsigned long long PL = (__ballot(pL) >> (grp * G)) & gmask;
    const unsigned long long zl = ~PL & below;
    const int runL = zl ? (gl - (63 - __clzll((long long)zl))) : (gl + 1);
    const int cH = pL ? (runL + ((runL == gl + 1) ? consH : 0)) : 0;
    const unsigned long long BH = (__ballot(act && cH > a.max_consecutive) >> (grp * G)) & gmask;
    const int kbH = BH ? (__ffsll((long long)BH) - 1) : G;
    // lower bound TL: stop on a run of possible collisions
    const unsigned long long PH = (__ballot(pH) >> (grp * G)) & gmask;
    const unsigned long long zh = ~PH & below;
    const int runH = zh ? (gl - (63 - __clzll((long long)zh))) : (gl + 1);
    const int cL = pH ? (runH + ((runH == gl + 1) ? consL : 0)) : 0;
    const unsigned long long BL = (__ballot(act && !doneL && cL > a.max_consecutive) >> (grp * G)) & gmask;
    const int kbL = BL ? (__ffsll((long long)BL) - 1) : G;
    // publish claims: every probe up to and including the terminating one
    if (act && gid != 0xFFFFFFFFu) {
      if (!a.l_only && gl <= kbH) claim_min(&a.ch_wr[gid], ch_val);
      if (!doneL && gl <= kbL && k >= tl_old) claim_min(&a.cl[gid], cl_val);
    }
    const int carryH = __shfl(cH, grp * G + (G - 1));
    const int carryL = __shfl(cL, grp * G + (G - 1));
    if (!done) {
      if (!doneL) {
        if (BL) { tl = base + kbL + 1; brokeL = true; doneL = true; }
        else { consL = carryL; if (base + G >= len) doneL = true; }
      }
      if (a.l_only) {
        done = doneL;
      } else {
        if (BH) { th = base + kbH + 1; brokeH = true; done = true; }
        else { consH = carryH; if (base + G >= len) done = true; }
      }
    }
  }
  bool open = false;
  if (gl == 0 && ray_ok) {
    a.TL[r] = tl;
    if (!a.l_only) a.TH[r] = th;
    const bool final_ray = !a.init && !a.l_only && (tl == th) && (brokeL == brokeH);
    if (final_ray) a.U[r] = brokeH ? th - 1 : th;  // the terminating probe's voxel is not updated
    else open = true;
  }
  return open;
}

// Appends the workgroup's open rays to the next sweep's work list: one global atomic per
// workgroup.  Every thread of the workgroup must call it.
__device__ inline void append_open(bool open, uint32_t r, uint32_t* list_out, uint32_t* counter) {
  __shared__ uint32_t s_cnt, s_base;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  uint32_t my = 0;
  if (open) my = atomicAdd(&s_cnt, 1u);
  __syncthreads();
  if (threadIdx.x == 0 && s_cnt) s_base = atomicAdd(counter, s_cnt);
  __syncthreads();
  if (open) list_out[s_base + my] = r;
  __syncthreads();
}

template <int G>
__global__ void __launch_bounds__(256) k_fast_sweep(SweepArgs a, uint32_t R, DevState* st) {
  const int lane = threadIdx.x & 63;
  const int grp = lane / G;
  const int gl = lane % G;
  constexpr int RPW = 64 / G;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t idx = wave * RPW + grp;
  const uint32_t n_in = a.list_in ? min(a.n_in, st->act_count[a.cnt_in]) : R;
  const bool ray_ok = idx < n_in;
  const uint32_t r = ray_ok ? (a.list_in ? a.list_in[idx] : idx) : 0;
  // the counter after the output one is the NEXT launch's output: zero it here
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->act_count[(a.cnt_out + 1) % 3] = 0;
    if (a.init) a.U[R] = 0;  // terminator of the exclusive scan over U
    if (a.list_in && n_in == 0) atomicMax(&st->fast_idle_sweep, 0xFFFFFFFFu - a.sweep_idx);
  }
  const bool open = sweep_ray<G, false>(a, ray_ok, r, grp, gl);
  if (a.list_out) append_open(open, r, a.list_out, &st->act_count[a.cnt_out]);
}

// ---------------------------------------------------------------------------
// Fast integrator, reference observed-voxel set (cfg.fast_observed_set == 0).
// voxel_observed_approx_set_ is an ApproxHashSet<20,10000> (tsdf_integrator.h:284-291): a probe
// of voxel v "collides" iff slot (hash(v) & 0xFFFFF) currently holds hash(v), i.e. iff the LATEST
// earlier probe of that slot had the same hash — voxels sharing a slot evict each other, so
// unlike the exact set the status of a voxel can flip back and the two-sided monotone solver
// above does not apply.  What still holds: a probe only depends on probes EARLIER in the
// 1-thread order (ray by ray, voxel by voxel).  So: guess every ray's probe count T (start: the
// exact-set solution), materialise all probes of the guess, order them by (slot, time) with one
// stable sort, read every probe's outcome off its predecessor in the slot, re-derive every
// ray's T from its outcomes, and repeat until no T moves.  At the fixed point every probe's
// outcome is consistent with all earlier probes, which by induction over time is the
// sequential execution.  One round = keys + 3-pass sort + two small kernels (~0.1 ms).
// ---------------------------------------------------------------------------
// key = slot(20) << 44 | hash bits 20..31 << 32 | probe index: the hash travels inside the key,
// so the sort moves 8 bytes per probe and no value array.  (Sorting only the upper 16 slot bits
// and walking back inside the 16-slot group was slower: groups next to the sensor hold
// thousands of probes of one hot slot.)
__global__ void k_strict_keys(const uint32_t* __restrict__ poff, uint32_t R, uint32_t P,
                              const uint32_t* __restrict__ off, const uint32_t* __restrict__ vox, MapDev m,
                              uint64_t* keys) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  uint32_t lo = 0, hi = R;  // largest r with poff[r] <= p
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (poff[mid] <= p) lo = mid; else hi = mid;
  }
  const uint32_t gid = vox[off[lo] + (p - poff[lo])];
  const uint32_t h = long_index_hash(voxel_of_gid(m, gid));
  keys[p] = ((uint64_t)(h & 0xFFFFFu) << 44) | ((uint64_t)(h >> 20) << 32) | p;  // p ascends in (ray, step) order = time
}
__device__ inline uint32_t strict_key_slot(uint64_t key) { return (uint32_t)(key >> 44); }
__device__ inline uint32_t strict_key_hash(uint64_t key) {
  return (uint32_t)(key >> 44) | ((uint32_t)((key >> 32) & 0xFFFu) << 20);
}
// replaceHash outcome of every probe (approx_hash_array.h:125-134): collision = the slot held
// this hash already.  set_vals = pseudo_set_ as the frame found it (at offset_).
__global__ void k_strict_outcome(const uint64_t* __restrict__ keys, uint32_t P,
                                 const uint32_t* __restrict__ set_vals, uint32_t offset, int sentinel_live,
                                 uint8_t* collided_by_p) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const uint64_t key = keys[i];
  const uint32_t slot = strict_key_slot(key);
  const uint32_t h = strict_key_hash(key);
  bool same;
  if (i > 0 && strict_key_slot(keys[i - 1]) == slot) {
    same = (strict_key_hash(keys[i - 1]) == h);
  } else {
    const uint32_t ai = slot + offset;
    same = !(ai == 0 && sentinel_live) && (set_vals[ai] == h);
  }
  collided_by_p[(uint32_t)(key & 0xFFFFFFFFu)] = same ? 1 : 0;
}
// Re-derives every ray's probe count from the outcomes of its guessed probes
// (tsdf_integrator.cc:531-551).  A ray whose guess ends before its walk does and that saw no
// terminating run must probe further: its guess grows and the next round tells.
__global__ void __launch_bounds__(256)
k_strict_scan(const uint32_t* __restrict__ poff, const uint32_t* __restrict__ off, uint32_t R,
              const uint8_t* __restrict__ collided, int max_consecutive, const uint32_t* __restrict__ T,
              uint32_t* Tnew, uint32_t* U, int debug_counts, DevState* st) {
  // 16 lanes per ray: 16 outcomes per step, the consecutive-collision counter is the run length
  // of the ballot mask (as in sweep_ray)
  constexpr int G = 16;
  const int lane = threadIdx.x & 63;
  const int grp = lane / G, gl = lane % G;
  const uint32_t r = (blockIdx.x * blockDim.x + threadIdx.x) / G;
  if (r == R && gl == 0) {
    Tnew[R] = 0;
    U[R] = 0;
  }
  const bool ray_ok = r < R;
  const uint32_t t = ray_ok ? T[r] : 0;
  const uint32_t len = ray_ok ? off[r + 1] - off[r] : 0;
  const uint32_t p0 = ray_ok ? poff[r] : 0;
  const unsigned long long gmask = (1ull << G) - 1ull;
  const unsigned long long below = (2ull << gl) - 1ull;
  int carry = 0;
  uint32_t tn = t;
  bool broke = false, done = (t == 0);
  for (uint32_t base = 0; __any(!done); base += G) {
    const uint32_t k = base + gl;
    const bool act = !done && k < t;
    const bool c = act && collided[p0 + k] != 0;
    const unsigned long long C = (__ballot(c) >> (grp * G)) & gmask;
    const unsigned long long z = ~C & below;
    const int run = z ? (gl - (63 - __clzll((long long)z))) : (gl + 1);
    const int cons = c ? (run + ((run == gl + 1) ? carry : 0)) : 0;
    const unsigned long long B = (__ballot(act && cons > max_consecutive) >> (grp * G)) & gmask;
    const int next_carry = __shfl(cons, grp * G + (G - 1));
    if (!done) {
      if (B) {
        tn = base + (uint32_t)(__ffsll((long long)B) - 1) + 1;
        broke = true;
        done = true;
      } else {
        carry = next_carry;
        if (base + G >= t) done = true;
      }
    }
  }
  if (gl == 0 && ray_ok) {
    if (!broke && t < len) tn = min(len, max(4u * t, t + 16u));  // surplus probes vanish again next round
    Tnew[r] = tn;
    U[r] = broke ? tn - 1 : tn;  // the terminating probe's voxel is not updated
    if (tn != t) {
      st->changed = 1;
      if (debug_counts) {  // same-address atomics serialise (~90/us): only on request (VBX_DEBUG)
        atomicAdd(&st->act_count[0], 1u);              // rays whose probe count moved this round
        if (!broke) atomicAdd(&st->act_count[1], 1u);  // of which: guesses that had to grow
      }
    }
  }
}
// The last probe of every slot leaves its hash in the persistent set.
__global__ void k_strict_commit(const uint64_t* __restrict__ keys, uint32_t P, uint32_t* set_vals, uint32_t offset,
                                DevState* st) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const uint64_t key = keys[i];
  const uint32_t slot = strict_key_slot(key);
  if (i + 1 < P && strict_key_slot(keys[i + 1]) == slot) return;  // a later probe of the same slot
  set_vals[slot + offset] = strict_key_hash(key);
  if (slot + offset == 0) st->sentinel_cleared = 1;
}

// clear_checks_every_n_frames > 1: the observed-voxel set outlives the frame, so every voxel a
// ray probed (k < T[r], the terminating probe included — it was inserted too,
// tsdf_integrator.cc:470-478) is stamped with the current epoch.  16 lanes per ray.
__global__ void k_fast_mark_observed(const uint32_t* __restrict__ off, const uint32_t* __restrict__ vox,
                                     const uint32_t* __restrict__ T, uint32_t R, uint32_t* obs, uint32_t epoch) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t r = t >> 4;
  if (r >= R) return;
  const uint32_t beg = off[r], len = T[r];
  for (uint32_t k = t & 15u; k < len; k += 16) {
    const uint32_t gid = vox[beg + k];
    if (gid != 0xFFFFFFFFu && obs[gid] != epoch) obs[gid] = epoch;
  }
}

// Emit the ordered update keys of the voxels each ray reaches (k < U[r]) straight from the
// voxel lists; one thread per key, the ray is found by binary search in the key offsets.
__global__ void k_fast_emit(const uint32_t* __restrict__ off_full, const uint32_t* __restrict__ vox,
                            const uint32_t* __restrict__ off_u, uint32_t R, uint32_t total,
                            MapDev m, uint64_t* keys, DevState* st) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  uint32_t lo = 0, hi = R;  // largest r with off_u[r] <= i
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (off_u[mid] <= i) lo = mid; else hi = mid;
  }
  const uint32_t r = lo;
  const uint32_t k = i - off_u[r];
  const uint32_t gid = vox[off_full[r] + k];
  if (gid == 0xFFFFFFFFu) {
    keys[i] = ~0ull;
    return;
  }
  keys[i] = ((uint64_t)gid << 32) | r;
  // tsdf_integrator.cc:128: block->updated().set() on every visited voxel's block
  const uint32_t slot = gid / m.nvox;
  const bool first_of_block = (k == 0) || (vox[off_full[r] + k - 1] / m.nvox != slot);
  if (first_of_block) publish_block(m, slot, st);
}



// ---------------------------------------------------------------------------
// kernels: Block<V>::serializeToIntegers / deserializeFromIntegers (src/core/block.cc)
// ---------------------------------------------------------------------------
__device__ inline uint32_t esdf_state_to_word(uint32_t st) {
  int px, py, pz;
  unpack_parent_bits(st, &px, &py, &pz);
  // serializeDirection (block.cc:8-40): int8 promoted to int, shifted, then cast to uint32 —
  // a negative component sign-extends over the higher bytes.
  uint32_t w = 0;
  w |= (uint32_t)((long long)(int8_t)px << 24);
  w |= (uint32_t)((long long)(int8_t)py << 16);
  w |= (uint32_t)((long long)(int8_t)pz << 8);
  w |= st & 0xFu;
  return w;
}
__global__ void k_serialize_tsdf(MapDev m, const uint32_t* __restrict__ slots, uint32_t* out) {
  const uint32_t slot = slots[blockIdx.x];
  if (slot == kInvalidSlot) return;
  uint32_t* o = out + (size_t)blockIdx.x * m.nvox * 3;
  for (uint32_t i = threadIdx.x; i < m.nvox * 3; i += blockDim.x) {  // coalesced word stream
    const uint32_t v = i / 3, f = i - 3 * v;
    const uint32_t gid = slot * m.nvox + v;
    uint32_t w;
    if (f == 0) w = __float_as_uint(m.dist[gid]);
    else if (f == 1) w = __float_as_uint(m.weight[gid]);
    else {
      const uint32_t c = m.rgba[gid];  // r | g<<8 | b<<16 | a<<24  ->  r<<24 | g<<16 | b<<8 | a
      w = ((c & 0xFF) << 24) | (((c >> 8) & 0xFF) << 16) | (((c >> 16) & 0xFF) << 8) | ((c >> 24) & 0xFF);
    }
    o[i] = w;
  }
}
__global__ void k_deserialize_tsdf(MapDev m, const uint32_t* __restrict__ slots, const uint32_t* __restrict__ in) {
  const uint32_t slot = slots[blockIdx.x];
  if (slot == kInvalidSlot) return;
  const uint32_t* w = in + (size_t)blockIdx.x * m.nvox * 3;
  for (uint32_t v = threadIdx.x; v < m.nvox; v += blockDim.x) {
    const uint32_t gid = slot * m.nvox + v;
    m.dist[gid] = __uint_as_float(w[3 * v]);
    m.weight[gid] = __uint_as_float(w[3 * v + 1]);
    const uint32_t c = w[3 * v + 2];
    m.rgba[gid] = ((c >> 24) & 0xFF) | (((c >> 16) & 0xFF) << 8) | (((c >> 8) & 0xFF) << 16) | ((c & 0xFF) << 24);
  }
}
__global__ void k_serialize_esdf(uint32_t nvox, const float* __restrict__ edist, const uint32_t* __restrict__ estate,
                                 const uint32_t* __restrict__ slots, uint32_t* out) {
  const uint32_t slot = slots[blockIdx.x];
  if (slot == kInvalidSlot) return;
  uint32_t* o = out + (size_t)blockIdx.x * nvox * 2;
  for (uint32_t i = threadIdx.x; i < nvox * 2; i += blockDim.x) {
    const uint32_t v = i >> 1;
    const uint32_t gid = slot * nvox + v;
    o[i] = (i & 1) ? esdf_state_to_word(estate[gid]) : __float_as_uint(edist[gid]);
  }
}
__global__ void k_deserialize_esdf(uint32_t nvox, float* edist, uint32_t* estate,
                                   const uint32_t* __restrict__ slots, const uint32_t* __restrict__ in) {
  const uint32_t slot = slots[blockIdx.x];
  if (slot == kInvalidSlot) return;
  const uint32_t* w = in + (size_t)blockIdx.x * nvox * 2;
  for (uint32_t v = threadIdx.x; v < nvox; v += blockDim.x) {
    const uint32_t gid = slot * nvox + v;
    edist[gid] = __uint_as_float(w[2 * v]);
    const uint32_t b = w[2 * v + 1];  // deserializeDirection (block.cc:42-64) + flag bits
    estate[gid] = (b & 0xFu) | (((b >> 24) & 0xFF) << 8) | (((b >> 16) & 0xFF) << 16) | (((b >> 8) & 0xFF) << 24);
  }
}
__global__ void k_set_block_flags(MapDev m, const uint32_t* __restrict__ slots, uint32_t n, uint32_t or_bits,
                                  const uint8_t* __restrict__ has_data, uint32_t has_data_bit) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || slots[i] == kInvalidSlot) return;
  uint32_t f = or_bits;
  if (has_data && has_data[i]) f |= has_data_bit;
  atomicOr(&m.blk_flags[slots[i]], f);
}

// ---------------------------------------------------------------------------
// kernels: multi-GPU block merge (mergeVoxelAIntoVoxelB as weighted sums)
// ---------------------------------------------------------------------------
__global__ void k_export_sums(MapDev m, const uint32_t* __restrict__ slots, float* out) {
  const uint32_t b = blockIdx.x;
  const uint32_t slot = slots[b];
  float* o = out + (size_t)b * 6 * m.nvox;
  for (uint32_t v = threadIdx.x; v < m.nvox; v += blockDim.x) {
    float wd = 0.f, w = 0.f, cr = 0.f, cg = 0.f, cb = 0.f, ca = 0.f;
    if (slot != kInvalidSlot) {
      const uint32_t gid = slot * m.nvox + v;
      w = m.weight[gid];
      wd = w * m.dist[gid];
      const uint32_t c = m.rgba[gid];
      cr = w * (float)(c & 0xFF); cg = w * (float)((c >> 8) & 0xFF);
      cb = w * (float)((c >> 16) & 0xFF); ca = w * (float)((c >> 24) & 0xFF);
    }
    o[v] = wd; o[m.nvox + v] = w; o[2 * m.nvox + v] = cr; o[3 * m.nvox + v] = cg;
    o[4 * m.nvox + v] = cb; o[5 * m.nvox + v] = ca;
  }
}

// lookup (find-only) of a host-provided block list -> slots; unpublished blocks read as absent
__global__ void k_lookup_slots(MapDev m, const int32_t* __restrict__ idx, uint32_t n, int published_only,
                               uint32_t* slots) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s = map_find(m, pack_block_key(idx[3 * i], idx[3 * i + 1], idx[3 * i + 2]));
  if (s != kInvalidSlot && published_only && !(m.blk_flags[s] & kFlagPublished)) s = kInvalidSlot;
  slots[i] = s;
}
__global__ void k_insert_blocks(MapDev m, const int32_t* __restrict__ idx, uint32_t n, uint32_t* new_list,
                                DevState* st) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  map_insert_key(m, pack_block_key(idx[3 * i], idx[3 * i + 1], idx[3 * i + 2]), new_list, st);
}

// Bulk mirror of blocks into the reference's AoS voxel layouts (voxel.h:12-37), one workgroup
// per requested block, coalesced word writes.  flags_out[b] = block flags, ~0u if the block is
// not part of the layer.
__global__ void k_lookup_slots_flags(MapDev m, const int32_t* __restrict__ idx, uint32_t n, uint32_t need,
                                     uint32_t* slots, uint32_t* flags_out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s = map_find(m, pack_block_key(idx[3 * i], idx[3 * i + 1], idx[3 * i + 2]));
  uint32_t f = ~0u;
  if (s != kInvalidSlot) {
    f = m.blk_flags[s];
    if (!(f & need)) { s = kInvalidSlot; f = ~0u; }
  }
  slots[i] = s;
  flags_out[i] = f;
}
__global__ void k_pack_tsdf_aos(MapDev m, const uint32_t* __restrict__ slots, uint32_t* out) {
  const uint32_t b = blockIdx.x;
  const uint32_t slot = slots[b];
  if (slot == kInvalidSlot) return;
  const uint32_t* d = reinterpret_cast<const uint32_t*>(m.dist) + (size_t)slot * m.nvox;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(m.weight) + (size_t)slot * m.nvox;
  const uint32_t* c = m.rgba + (size_t)slot * m.nvox;
  uint32_t* o = out + (size_t)b * m.nvox * 3;
  for (uint32_t i = threadIdx.x; i < m.nvox * 3; i += blockDim.x) {
    const uint32_t v = i / 3, k = i % 3;
    o[i] = (k == 0) ? d[v] : (k == 1 ? w[v] : c[v]);  // {float distance; float weight; Color color}
  }
}
__global__ void k_pack_esdf_aos(uint32_t nvox, const float* __restrict__ edist, const uint32_t* __restrict__ estate,
                                const uint32_t* __restrict__ slots, uint32_t* out) {
  const uint32_t b = blockIdx.x;
  const uint32_t slot = slots[b];
  if (slot == kInvalidSlot) return;
  const uint32_t* d = reinterpret_cast<const uint32_t*>(edist) + (size_t)slot * nvox;
  const uint32_t* st = estate + (size_t)slot * nvox;
  uint32_t* o = out + (size_t)b * nvox * 5;
  for (uint32_t i = threadIdx.x; i < nvox * 5; i += blockDim.x) {
    const uint32_t v = i / 5, k = i % 5;
    uint32_t wv;
    if (k == 0) {
      wv = d[v];
    } else {
      const uint32_t x = st[v];
      if (k == 1)  // bool observed, hallucinated, in_queue, fixed: one byte each
        wv = (x & 1u) | ((x & 2u) << 7) | ((x & 4u) << 14) | ((x & 8u) << 21);
      else         // Eigen::Vector3i parent
        wv = (uint32_t)(int32_t)(int8_t)((x >> (8 * (k - 1))) & 0xFFu);
    }
    o[i] = wv;
  }
}

__global__ void k_merge_sums(MapDev m, const uint32_t* __restrict__ slots, const float* __restrict__ in,
                             int apply_caps, float trunc, float max_weight, DevState* st) {
  const uint32_t b = blockIdx.x;
  const uint32_t slot = slots[b];
  if (slot == kInvalidSlot) return;
  const float* a = in + (size_t)b * 6 * m.nvox;
  bool any = false;
  for (uint32_t v = threadIdx.x; v < m.nvox; v += blockDim.x) {
    const float wA = a[m.nvox + v];
    if (!(wA > 0.0f)) continue;
    any = true;
    const uint32_t gid = slot * m.nvox + v;
    const float dA = a[v] / wA;
    uint32_t cA = 0;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      const float c = roundf(a[(2 + ch) * m.nvox + v] / wA);
      cA |= ((uint32_t)(int)std_min(std_max(c, 0.0f), 255.0f) & 0xFFu) << (8 * ch);
    }
    const float wB = m.weight[gid];
    const float dB = m.dist[gid];
    const float cw = wA + wB;  // mergeVoxelAIntoVoxelB, voxel_utils.cc:10-22
    if (cw > 0.0f) {
      float d = (dA * wA + dB * wB) / cw;
      float w = cw;
      const uint32_t col = blend_two_colors(cA, wA, m.rgba[gid], wB);
      if (apply_caps) {
        d = (d > 0.0f) ? std_min(trunc, d) : std_max(-trunc, d);
        w = std_min(max_weight, w);
      }
      m.dist[gid] = d;
      m.weight[gid] = w;
      m.rgba[gid] = col;
    }
  }
  if (__syncthreads_or(any ? 1 : 0) && threadIdx.x == 0) publish_block(m, slot, st);
}

// Layer::removeDistantBlocks (layer.h:170-182) for every block of one layer in one launch: a
// workgroup per pool slot; (origin - center).squaredNorm() > max^2 with origin = float(index) *
// block_size (common.h:195-201).  A removed block is zeroed and leaves the layer; its hash
// entry and pool slot stay (an invisible candidate again).
__global__ void k_remove_distant(MapDev m, float* edist, uint32_t* estate, int layer, f3 center, double max_sq,
                                 float block_size) {
  const uint32_t slot = blockIdx.x;
  const uint32_t f = m.blk_flags[slot];
  const uint32_t need = (layer == VBX_LAYER_ESDF) ? kFlagEsdfAlloc : kFlagPublished;
  if (!(f & need)) return;
  const f3 o{(float)m.blk_idx[3 * slot] * block_size, (float)m.blk_idx[3 * slot + 1] * block_size,
             (float)m.blk_idx[3 * slot + 2] * block_size};
  if (!((double)f3_sqnorm(f3_sub(o, center)) > max_sq)) return;
  const size_t base = (size_t)slot * m.nvox;
  if (layer == VBX_LAYER_ESDF) {
    if (edist)
      for (uint32_t v = threadIdx.x; v < m.nvox; v += blockDim.x) { edist[base + v] = 0.f; estate[base + v] = 0u; }
  } else {
    for (uint32_t v = threadIdx.x; v < m.nvox; v += blockDim.x) {
      m.dist[base + v] = 0.f; m.weight[base + v] = 0.f; m.rgba[base + v] = 0u;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // the two layers are independent (layer.h:167): keep the other layer's membership and bits
    const uint32_t esdf_bits = kFlagEsdfAlloc | (kFlagUpdMask << kFlagEsdfUpdShift) | kFlagEsdfPendClassify | kFlagEsdfPendOpen;
    m.blk_flags[slot] = (layer == VBX_LAYER_ESDF) ? (f & ~esdf_bits) : (f & esdf_bits);
  }
}

// Block::updated().reset(bits) on every block of one layer
__global__ void k_clear_update_bits(MapDev m, uint32_t n_slots, uint32_t need, uint32_t bits) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  const uint32_t f = m.blk_flags[s];
  if ((f & need) && (f & bits)) m.blk_flags[s] = f & ~bits;
}
__global__ void k_reset_tsdf_flags(MapDev m, uint32_t n_slots) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  m.blk_flags[s] &= (kFlagEsdfAlloc | (kFlagUpdMask << kFlagEsdfUpdShift));
}

// ===========================================================================
// ESDF integrator (esdf_integrator.cc) — see DESIGN.md §ESDF.
//
// The reference runs three strictly sequential phases per update: (1) walk every voxel of the
// updated TSDF blocks and classify it (new / lower / raise / sign flip), (2) a FIFO "raise"
// wavefront that invalidates the children (by parent pointer) of voxels whose distance grew,
// (3) a bucket-queue "lower" wavefront that relaxes 26-neighbours until nothing improves.
// Phase 1 is per-voxel independent and is reproduced rule by rule.  Phases 2 and 3 compute
// closures / fixed points that do not depend on the visiting order (for min_diff_m == 0 the
// lower phase is Bellman-Ford on the 26-graph: every voxel ends at the float-minimum over
// all paths), so they run as LDS-tiled chaotic relaxations: one workgroup stages a block plus
// its one-voxel halo (18^3 distances + states, 46 KiB) in LDS, relaxes it to a local fixed
// point, and the host repeats global sweeps until no block changes.
// ===========================================================================
constexpr uint32_t kEsdfObserved = 1, kEsdfHallucinated = 2, kEsdfInQueue = 4, kEsdfFixed = 8;

struct EsdfDev {
  float* dist;
  uint32_t* state;    // bits 0-3 flags, 8-15 / 16-23 / 24-31 parent x/y/z (int8)
  uint8_t* raised;    // 1 = raised during the current update
  uint32_t* active;   // per slot: 1 = process this sweep, 2 = process next sweep, 4 = touched
};
struct EsdfCfgDev {
  float max_distance, min_distance, default_distance, min_diff, min_weight;
  int add_occupied_crust;
  float voxel_size;
};

constexpr int kNbOff[26][3] = {
    {-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1},
    {-1, -1, 0}, {-1, 1, 0}, {1, -1, 0}, {1, 1, 0}, {0, -1, -1}, {0, -1, 1},
    {0, 1, -1}, {0, 1, 1}, {-1, 0, -1}, {1, 0, -1}, {-1, 0, 1}, {1, 0, 1},
    {-1, -1, -1}, {-1, -1, 1}, {-1, 1, -1}, {-1, 1, 1}, {1, -1, -1}, {1, -1, 1},
    {1, 1, -1}, {1, 1, 1}};  // same table, compile-time (unrolled loops)
__constant__ int c_nb_off[26][3] = {
    {-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1},
    {-1, -1, 0}, {-1, 1, 0}, {1, -1, 0}, {1, 1, 0}, {0, -1, -1}, {0, -1, 1},
    {0, 1, -1}, {0, 1, 1}, {-1, 0, -1}, {1, 0, -1}, {-1, 0, 1}, {1, 0, 1},
    {-1, -1, -1}, {-1, -1, 1}, {-1, 1, -1}, {-1, 1, 1}, {1, -1, -1}, {1, -1, 1},
    {1, 1, -1}, {1, 1, 1}};  // neighbor_tools.cc:24-30, column order is observable

__device__ inline uint32_t pack_parent(int x, int y, int z) {
  return ((uint32_t)(uint8_t)(int8_t)x << 8) | ((uint32_t)(uint8_t)(int8_t)y << 16) |
         ((uint32_t)(uint8_t)(int8_t)z << 24);
}
__device__ inline void unpack_parent(uint32_t s, int* x, int* y, int* z) {
  *x = (int)(int8_t)((s >> 8) & 0xFF);
  *y = (int)(int8_t)((s >> 16) & 0xFF);
  *z = (int)(int8_t)((s >> 24) & 0xFF);
}

// Phase 1: EsdfIntegrator::updateFromTsdfBlocks, esdf_integrator.cc:136-287, one thread per
// voxel of every TSDF block that carries the kEsdf update bit (incremental) or of every
// allocated TSDF block (batch).  Queue pushes become marks: `raised` for raise_.push, block
// activity for open_.push (the lower phase re-relaxes whole active blocks).
// updateVoxelFromNeighbors (:498-530) is a pull from already-converged neighbours; the lower
// phase's pull relaxation subsumes it (and does not reproduce its unscaled-distance quirk).
// select: 0 = every TSDF block (batch), 1 = blocks with Update::kEsdf or in updated_blocks_
// (updateFromTsdfLayer), 2 = only the blocks the caller listed (updateFromTsdfBlocks).
__global__ void k_esdf_classify(MapDev m, EsdfDev e, EsdfCfgDev c, int incremental, int select, DevState* st) {
  const uint32_t slot = blockIdx.x;
  const uint32_t flags = m.blk_flags[slot];
  if (!(flags & kFlagPublished)) return;
  // Update::kEsdf, or a member of updated_blocks_ (esdf_integrator.cc:107-108)
  if (select == 1 && !(flags & 4u) && !(e.active[slot] & 16u)) return;
  if (select == 2 && !(e.active[slot] & 32u)) return;
  const uint32_t lin = blockIdx.y * blockDim.x + threadIdx.x;
  if (lin >= m.nvox) return;
  if (lin == 0) {
    atomicOr(&m.blk_flags[slot], kFlagEsdfAlloc | (1u << kFlagEsdfUpdShift));  // set_updated(true): kMap only
    atomicOr(&e.active[slot], 8u);  // processed by this update
    atomicAdd(&st->esdf_blocks, 1u);
  }
  const uint32_t gid = slot * m.nvox + lin;
  const float td = m.dist[gid];
  const float tw = m.weight[gid];
  float ed = e.dist[gid];
  uint32_t es = e.state[gid];
  if (tw < c.min_weight) {
    if (!incremental && c.add_occupied_crust) {
      ed = -c.default_distance;
      es = (es | kEsdfObserved | kEsdfHallucinated) & ~kEsdfFixed;
      e.dist[gid] = ed;
      e.state[gid] = es;
    }
    return;
  }
  const bool tsdf_fixed = fabsf(td) < c.min_distance;
  const float sgn_default = (float)signum(td) * c.default_distance;
  bool raise = false;
  if (!(es & kEsdfObserved) || (es & kEsdfHallucinated)) {
    if (es & kEsdfHallucinated) raise = true;
    if (tsdf_fixed) {
      ed = td;
      es |= kEsdfFixed;
    } else {
      ed = sgn_default;
      es &= ~kEsdfFixed;
    }
    es &= 0xFFu;  // parent.setZero()
  } else {
    const bool efixed = (es & kEsdfFixed) != 0;
    if (tsdf_fixed || efixed) {
      if (!tsdf_fixed) {
        ed = sgn_default;
        es &= 0xFFu;
        es &= ~kEsdfFixed;
        raise = true;
      } else if ((ed > 0.0f && td + c.min_diff < ed) || (ed <= 0.0f && td - c.min_diff > ed)) {
        es |= kEsdfFixed;  // fixed = tsdf_fixed (true here)
        ed = td;
        es &= 0xFFu;
      } else if ((ed > 0.0f && td - c.min_diff > ed) || (ed <= 0.0f && td + c.min_diff < ed)) {
        es |= kEsdfFixed;
        ed = td;
        es &= 0xFFu;
        raise = true;
      }
    } else if (signum(td) != signum(ed)) {
      if (td < ed) {
        ed = sgn_default;
        es &= 0xFFu;
      } else {
        ed = sgn_default;
        es &= 0xFFu;
        raise = true;
      }
    }
  }
  es |= kEsdfObserved;
  es &= ~(kEsdfHallucinated | kEsdfInQueue);
  e.dist[gid] = ed;
  e.state[gid] = es;
  if (raise) {
    e.raised[gid] = 1;
    st->esdf_raise_any = 1;
  }
}

// ---------------------------------------------------------------------------
// EsdfIntegrator::addNewRobotPosition, esdf_integrator.cc:25-92, over the sphere voxel lists of
// utils::getSphereAroundPoint (planning_utils_inl.h:14-48).  `xs` holds the reference's float
// loop variable (x = -r; x <= r; x++) computed on the host with the same increments; one thread
// per (i,j,k) of the n^3 cube around the centre voxel.
// ---------------------------------------------------------------------------
struct SphereDev {
  const float* xs;
  int n;
  float r;      // radius in voxels
  l3 center;    // getGridIndexFromPoint<GlobalIndex>(center, voxel_size_inv)
};
__device__ inline bool sphere_voxel(const SphereDev& sp, const MapDev& m, size_t t, uint64_t* key, uint32_t* lin) {
  const size_t n = (size_t)sp.n;
  if (t >= n * n * n) return false;
  const int k = (int)(t % n), j = (int)((t / n) % n), i = (int)(t / (n * n));
  const f3 pv{sp.xs[i], sp.xs[j], sp.xs[k]};
  if (!(f3_norm(pv) <= sp.r)) return false;
  const l3 g{(int64_t)floorf(pv.x) + sp.center.x, (int64_t)floorf(pv.y) + sp.center.y,
             (int64_t)floorf(pv.z) + sp.center.z};
  const i3 b = block_index_from_global(g, m.vps_inv);  // common.h:245-255
  const i3 v = local_from_global(g, (int)m.vps);
  *key = pack_block_key(b.x, b.y, b.z);
  *lin = (uint32_t)v.x + m.vps * ((uint32_t)v.y + (uint32_t)v.z * m.vps);
  return true;
}
// getAndAllocateSphereAroundPoint (planning_utils_inl.h:50-61): every block holding a sphere voxel
__global__ void k_sphere_mark_blocks(MapDev m, SphereDev sp, uint32_t* new_list, DevState* st) {
  uint64_t key;
  uint32_t lin;
  if (!sphere_voxel(sp, m, (size_t)blockIdx.x * blockDim.x + threadIdx.x, &key, &lin)) return;
  map_insert_key(m, key, new_list, st);
}
// mode 0: inner sphere (unknown or hallucinated -> free); mode 1: outer sphere (unknown ->
// occupied, known -> open_).  Queue pushes become marks that the next update consumes.
__global__ void k_sphere_apply(MapDev m, EsdfDev e, SphereDev sp, float default_distance, int mode) {
  uint64_t key;
  uint32_t lin;
  if (!sphere_voxel(sp, m, (size_t)blockIdx.x * blockDim.x + threadIdx.x, &key, &lin)) return;
  const uint32_t slot = map_find(m, key);
  if (slot == kInvalidSlot) return;  // pool exhausted: reported through DevState::error
  const uint32_t gid = slot * m.nvox + lin;
  uint32_t want = kFlagEsdfAlloc;
  const uint32_t es = e.state[gid];
  if (mode == 0) {
    if (!(es & kEsdfObserved) || (es & kEsdfHallucinated)) {
      if (es & kEsdfHallucinated) e.raised[gid] = 1;  // raise_.push
      e.dist[gid] = default_distance;
      e.state[gid] = (es & 0xFFu) | kEsdfObserved | kEsdfHallucinated;  // parent.setZero()
      want |= kFlagEsdfPendClassify;
    }
  } else {
    if (!(es & kEsdfObserved)) {
      e.dist[gid] = -default_distance;
      e.state[gid] = (es & 0xFFu) | kEsdfObserved | kEsdfHallucinated;
      want |= kFlagEsdfPendClassify;
    } else {
      want |= kFlagEsdfPendOpen;  // open_.push(global_index, distance) — in_queue stays clear (:81-85)
    }
  }
  const uint32_t cur = __hip_atomic_load(&m.blk_flags[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((cur & want) != want) atomicOr(&m.blk_flags[slot], want);
}

// updateFromTsdfBlocks(list): mark the listed blocks for classification
__global__ void k_esdf_mark_listed(MapDev m, EsdfDev e, const int32_t* __restrict__ idx, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t s = map_find(m, pack_block_key(idx[3 * i], idx[3 * i + 1], idx[3 * i + 2]));
  if (s != kInvalidSlot) atomicOr(&e.active[s], 32u);
}

// active(cur) = every block processed by this update and its 26 neighbours.
__global__ void k_esdf_seed_active(MapDev m, EsdfDev e, uint32_t n_slots) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t slot = i / 27, nb = i % 27;
  if (slot >= n_slots) return;
  if (!(e.active[slot] & 8u)) return;
  const int dx = (int)(nb % 3) - 1, dy = (int)((nb / 3) % 3) - 1, dz = (int)(nb / 9) - 1;
  const uint32_t s2 = map_find(m, pack_block_key(m.blk_idx[3 * slot] + dx, m.blk_idx[3 * slot + 1] + dy,
                                                 m.blk_idx[3 * slot + 2] + dz));
  if (s2 != kInvalidSlot && (m.blk_flags[s2] & kFlagEsdfAlloc)) atomicOr(&e.active[s2], 1u | 4u);
}
__global__ void k_esdf_rotate_active(EsdfDev e, uint32_t n_slots, int reseed) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  const uint32_t a = e.active[s];
  uint32_t cur = (a & 2u) ? 1u : 0u;
  if (reseed) cur = (a & 4u) ? 1u : 0u;  // start of a new phase: everything touched so far
  e.active[s] = cur | (a & 12u) | (cur ? 4u : 0u);
}

// Phases 2/3 (+ parent canonicalisation) on one block + halo staged in LDS.
//   mode 0: processRaiseSet (esdf_integrator.cc:305-369) as a closure: a non-fixed observed
//           voxel whose parent voxel was raised is reset to sign*default and raised itself.
//   mode 1: processOpenSet (:371-496) as a pull relaxation over the 26-neighbourhood.
//   mode 2: parent = first LUT neighbour that explains the converged distance exactly.
constexpr int kEsdfThreads = 1024;  // one workgroup relaxes one block; big frontiers need the lanes
template <int VPS>
__global__ void __launch_bounds__(kEsdfThreads) k_esdf_tile(MapDev m, EsdfDev e, EsdfCfgDev c, int mode,
                                                   DevState* st) {
  constexpr int T = VPS + 2;
  constexpr int NT = T * T * T;
  constexpr int NV = VPS * VPS * VPS;
  __shared__ float s_d[NT];
  __shared__ uint32_t s_s[NT];
  __shared__ uint8_t s_r[NT];     // raise marks (mode 0) / need flags (mode 1): never both
  __shared__ uint16_t s_q[NV];    // mode 1: dense work queue
  __shared__ int s_qn;
  __shared__ uint32_t s_nb[27];
  __shared__ int s_flag;
  const uint32_t slot = blockIdx.x;
  if (!(m.blk_flags[slot] & kFlagEsdfAlloc)) return;
  if (!(e.active[slot] & 1u)) return;
  const int tid = threadIdx.x;
  if (tid < 27) {
    const int dx = tid % 3 - 1, dy = (tid / 3) % 3 - 1, dz = tid / 9 - 1;
    uint32_t s2 = map_find(m, pack_block_key(m.blk_idx[3 * slot] + dx, m.blk_idx[3 * slot + 1] + dy,
                                             m.blk_idx[3 * slot + 2] + dz));
    if (s2 != kInvalidSlot && !(m.blk_flags[s2] & kFlagEsdfAlloc)) s2 = kInvalidSlot;
    s_nb[tid] = s2;
  }
  if (tid == 0) s_flag = 0;
  __syncthreads();
#pragma unroll 4
  for (int t = tid; t < NT; t += kEsdfThreads) {
    const int tx = t % T, ty = (t / T) % T, tz = t / (T * T);
    const int bx = (tx == 0) ? 0 : (tx == T - 1 ? 2 : 1);
    const int by = (ty == 0) ? 0 : (ty == T - 1 ? 2 : 1);
    const int bz = (tz == 0) ? 0 : (tz == T - 1 ? 2 : 1);
    const uint32_t s2 = s_nb[bx + 3 * by + 9 * bz];
    float d = 0.f;
    uint32_t s = 0;  // getVoxelPtrByGlobalIndex == nullptr: looks unobserved
    uint8_t r = 0;
    if (s2 != kInvalidSlot) {
      const int lx = (tx - 1) & (VPS - 1), ly = (ty - 1) & (VPS - 1), lz = (tz - 1) & (VPS - 1);
      const uint32_t g2 = s2 * NV + (uint32_t)(lx + VPS * (ly + lz * VPS));
      d = e.dist[g2];
      s = e.state[g2];
      r = e.raised[g2];
    }
    s_d[t] = d;
    s_s[t] = s;
    s_r[t] = r;
  }
  __syncthreads();

  const float sq2 = (float)1.4142135623730951, sq3 = (float)1.7320508075688772;
  bool any_change = false;

  // processOpenSet for one voxel (pull form): returns true if the voxel was lowered.
  auto relax = [&](int t) -> bool {
    const uint32_t s = s_s[t];
    if (!(s & kEsdfObserved) || (s & kEsdfFixed)) return false;
    float d = s_d[t];
    bool upd = false;
    int best = -1;
    // fully unrolled: the 52 LDS reads of one voxel issue back to back
#pragma unroll
    for (int i = 0; i < 26; ++i) {
      const int tv = t + kNbOff[i][0] + T * (kNbOff[i][1] + T * kNbOff[i][2]);
      const uint32_t sv = s_s[tv];
      if (!(sv & kEsdfObserved)) continue;
      const float dv = s_d[tv];
      if (dv >= c.max_distance || dv <= -c.max_distance) continue;
      const float dist = (i < 6 ? 1.0f : (i < 18 ? sq2 : sq3)) * c.voxel_size;
      if (dv > 0 && d > 0) {
        if (dv + dist + c.min_diff < d) { d = dv + dist; best = i; upd = true; }
      } else if (dv <= 0 && d <= 0) {
        if (dv - dist - c.min_diff > d) { d = dv - dist; best = i; upd = true; }
      } else {
        // sign mismatch (esdf_integrator.cc:459-488).  In the reference this assignment is
        // gated by |potential - d| > dist and its outcome depends on the pop order of the two
        // neighbours (libstdc++ unordered_map block order).  The order-free form used here
        // applies the same candidate whenever it moves the voxel closer to the surface, which
        // is the outcome of the reference when the opposite-sign neighbour pops first.
        const float potential = dv - (float)signum(dv) * dist;
        float cand;
        if ((float)signum(potential) == d) cand = potential;
        else cand = (float)signum(d) * dist;
        if (fabsf(cand) < fabsf(d)) { d = cand; best = i; upd = true; }
      }
    }
    if (upd) {
      s_d[t] = d;
      s_s[t] = (s & 0xFFu) | pack_parent(c_nb_off[best][0], c_nb_off[best][1], c_nb_off[best][2]);
    }
    return upd;
  };

  if (mode == 1) {
    // Worklist relaxation: s_need marks voxels whose neighbourhood changed; every iteration
    // compacts the marked voxels into a dense queue (so all lanes evaluate real work), relaxes
    // them, and marks the 26 neighbours of every voxel that moved.  Total evaluations are
    // proportional to the number of changes, not to iterations x block size.  (A push-based
    // queue with an atomic visited bitset was measured slower: 3.1 vs 2.2 ms per update.)
    uint8_t* s_need = s_r;  // the raise marks are not used while lowering
    for (int t = tid; t < NT; t += kEsdfThreads) s_need[t] = 0;
    __syncthreads();
    for (int v = tid; v < NV; v += kEsdfThreads) {
      const int lx = v % VPS, ly = (v / VPS) % VPS, lz = v / (VPS * VPS);
      s_need[(lx + 1) + T * ((ly + 1) + T * (lz + 1))] = 1;  // first pass: everything
    }
    for (int iter = 0; iter < 64 * VPS; ++iter) {
      if (tid == 0) s_qn = 0;
      __syncthreads();
      for (int v = tid; v < NV; v += kEsdfThreads) {
        const int lx = v % VPS, ly = (v / VPS) % VPS, lz = v / (VPS * VPS);
        const int t = (lx + 1) + T * ((ly + 1) + T * (lz + 1));
        if (s_need[t]) {
          s_need[t] = 0;
          const uint32_t sv = s_s[t];
          if ((sv & kEsdfObserved) && !(sv & kEsdfFixed)) s_q[atomicAdd(&s_qn, 1)] = (uint16_t)t;
        }
      }
      __syncthreads();
      const int qn = s_qn;
      if (qn == 0) break;
      for (int q = tid; q < qn; q += kEsdfThreads) {
        const int t = s_q[q];
        if (relax(t)) {
          any_change = true;
#pragma unroll
          for (int i = 0; i < 26; ++i) s_need[t + kNbOff[i][0] + T * (kNbOff[i][1] + T * kNbOff[i][2])] = 1;
        }
      }
      __syncthreads();
    }
  } else {
  for (int iter = 0; iter < 4 * VPS; ++iter) {
    bool changed = false;
    for (int v = tid; v < NV; v += kEsdfThreads) {
      const int lx = v % VPS, ly = (v / VPS) % VPS, lz = v / (VPS * VPS);
      const int t = (lx + 1) + T * ((ly + 1) + T * (lz + 1));
      uint32_t s = s_s[t];
      if (!(s & kEsdfObserved) || (s & kEsdfFixed)) continue;
      float d = s_d[t];
      if (mode == 0) {
        int px, py, pz;
        unpack_parent(s, &px, &py, &pz);
        if ((px | py | pz) == 0 || s_r[t]) continue;
        // quasi-Euclidean parents are unit LUT offsets, so the parent voxel is inside the halo
        const int tp = t + px + T * (py + T * pz);
        if (s_r[tp]) {
          s_d[t] = (float)signum(d) * c.default_distance;
          s_s[t] = s & 0xFFu;
          s_r[t] = 1;
          changed = true;
        }
        continue;
      }
      // mode 2: canonical parent
      {
        int px, py, pz;
        unpack_parent(s, &px, &py, &pz);
        if ((px | py | pz) == 0) continue;
        for (int i = 0; i < 26; ++i) {
          const int tv = t + c_nb_off[i][0] + T * (c_nb_off[i][1] + T * c_nb_off[i][2]);
          const uint32_t sv = s_s[tv];
          if (!(sv & kEsdfObserved)) continue;
          const float dv = s_d[tv];
          if (dv >= c.max_distance || dv <= -c.max_distance) continue;
          const float dist = (i < 6 ? 1.0f : (i < 18 ? sq2 : sq3)) * c.voxel_size;
          bool hit;
          if (dv > 0 && d > 0) hit = (dv + dist == d);
          else if (dv <= 0 && d <= 0) hit = (dv - dist == d);
          else {  // the sign-mismatch rule: d == potential or sign(d) * dist
            const float potential = dv - (float)signum(dv) * dist;
            const float cand = ((float)signum(potential) == d) ? potential : (float)signum(d) * dist;
            hit = (cand == d);
          }
          if (hit) {
            const uint32_t ns = (s & 0xFFu) | pack_parent(c_nb_off[i][0], c_nb_off[i][1], c_nb_off[i][2]);
            if (ns != s) { s_s[t] = ns; changed = true; }
            break;
          }
        }
      }
    }
    any_change |= changed;
    const int more = __syncthreads_or(changed ? 1 : 0);
    if (!more || mode == 2) break;
  }
  }
  if (any_change) s_flag = 1;
  __syncthreads();
  if (!s_flag) return;
  for (int v = tid; v < NV; v += kEsdfThreads) {
    const int lx = v % VPS, ly = (v / VPS) % VPS, lz = v / (VPS * VPS);
    const int t = (lx + 1) + T * ((ly + 1) + T * (lz + 1));
    const uint32_t g = slot * NV + v;
    e.dist[g] = s_d[t];
    e.state[g] = s_s[t];
    if (mode == 0) e.raised[g] = s_r[t];
  }
  if (mode != 2) {
    if (tid < 27 && s_nb[tid] != kInvalidSlot) atomicOr(&e.active[s_nb[tid]], 2u | 4u);
    if (tid == 0) {
      st->changed = 1;
      atomicAdd(&st->esdf_relax_blocks, 1u);
    }
  }
}

}  // namespace

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
struct vbx_ctx {
  int device = 0;
  vbx_map_cfg mcfg{};
  MapDev map{};
  uint32_t hcap = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  DevState* d_state = nullptr;
  DevState h_state{};
  StateMirror* h_mirror = nullptr;  // page-locked, mapped (may stay null: plain copies are used then)
  StateMirror* d_mirror = nullptr;
  uint32_t sync_seq = 0;
  std::string err;

  // pool / map storage
  DBuf b_hkeys, b_hvals, b_dist, b_weight, b_rgba, b_blkidx, b_blkflags, b_freelist, b_newlist;
  // per-call scratch
  DBuf b_pts, b_cols;                                   // staged host input
  DBuf t_px, t_py, t_pz, t_rgba, t_w, t_flags, t_bkey;  // ray table A (per point, order s)
  DBuf u_px, u_py, u_pz, u_rgba, u_w, u_flags, u_bkey;  // ray table B (bundles / kept rays)
  DBuf b_pcx, b_pcy, b_pcz;                             // Merged: point_C per s
  DBuf b_cnt, b_off, b_keys0, b_keys1, b_vals0, b_vals1, b_tmp, b_head, b_rank, b_graze;
  DBuf b_T, b_TH, b_U, b_vox, b_cl, b_act0, b_act1, b_long, b_order, b_obs, b_sphere0, b_sphere1, b_redo, b_bkeys, b_bfirst, b_bperm, b_obsset, b_collided;
  uint32_t obs_epoch = 1;
  uint32_t fast_last_iters = 0;  // sweeps the previous Fast frame needed
  uint32_t fast_redo_grid = 0;   // rays the second list-building pass is launched for
  // Fast integrator persistent state
  DBuf b_startset;       // ApproxHashSet<20,10000> storage (u32 per slot)
  uint32_t start_offset = 0;
  bool start_sentinel_live = true;
  bool startset_init = false;
  std::vector<int32_t> h_by_s;  // scratch of merged_reference_order
  // voxel_observed_approx_set_ (reference semantics, fast_observed_set == 0)
  uint32_t obsset_offset = 0;
  bool obsset_sentinel_live = true;
  bool obsset_init = false;
  int64_t reset_counter = 0;  // tsdf_integrator.cc:564
  DBuf b_own0, b_own1;
  // ESDF layer (allocated on first use)
  DBuf b_edist, b_estate, b_eraised, b_eactive;
  bool esdf_init = false;
  bool esdf_robot_pending = false;  // addNewRobotPosition since the last update
  uint32_t own_tag = 0;  // descending
  int own_s_bits = 0;

  vbx_counters counters{};
  bool timing = false;
  hipEvent_t ev[9] = {};  // 0..7 stage boundaries in order, 8 = end of the exact-set solve inside stage 3
  bool ev_hit[9] = {};
  vbx_timing last_timing{};

  void fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    err = buf;
  }
};

namespace {

inline dim3 grid_for(size_t n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }

// Reads DevState (and optionally one more device word) back to the host; on return everything
// queued on the stream before the call has completed.
int sync_state(vbx_ctx* ctx, const uint32_t* d_extra = nullptr, uint32_t* extra_out = nullptr) {
  if (!ctx->h_mirror) {
    HIP_TRY(hipMemcpyAsync(&ctx->h_state, ctx->d_state, sizeof(DevState), hipMemcpyDeviceToHost, ctx->stream));
    if (d_extra) HIP_TRY(hipMemcpyAsync(extra_out, d_extra, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VBX_OK;
  }
  const uint32_t seq = ++ctx->sync_seq;
  hipLaunchKernelGGL(k_publish_state, dim3(1), dim3(64), 0, ctx->stream, ctx->d_state, ctx->d_mirror, d_extra, seq);
  HIP_TRY(hipGetLastError());
  const auto t0 = std::chrono::steady_clock::now();
  uint32_t spins = 0;
  while (__atomic_load_n(&ctx->h_mirror->seq, __ATOMIC_ACQUIRE) != seq) {
    if ((++spins & 0xFFFu) == 0 &&
        std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
      // long-running or failed work: block, and let the runtime report an error if there is one
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      if (__atomic_load_n(&ctx->h_mirror->seq, __ATOMIC_ACQUIRE) != seq) {
        ctx->fail("device state read-back did not arrive");
        return VBX_ERR_HIP;
      }
      break;
    }
  }
  std::memcpy(&ctx->h_state, &ctx->h_mirror->st, sizeof(DevState));
  if (d_extra) *extra_out = ctx->h_mirror->extra;
  return VBX_OK;
}

int check_state_error(vbx_ctx* ctx) {
  if (ctx->h_state.error & 1u) {
    ctx->fail("block pool / hash map capacity exceeded (max_blocks=%u)", ctx->map.cap_blocks);
    return VBX_ERR_CAPACITY;
  }
  if (ctx->h_state.error & 2u) {
    ctx->fail("internal: ray march hit a block without a pool slot");
    return VBX_ERR_HIP;
  }
  if (ctx->h_state.error & 4u) {
    ctx->fail("internal: voxel list capacity bound violated");
    return VBX_ERR_HIP;
  }
  return VBX_OK;
}

RayTab make_tab(vbx_ctx* ctx, bool second, uint32_t R) {
  RayTab t;
  if (!second) {
    t.px = ctx->t_px.as<float>(); t.py = ctx->t_py.as<float>(); t.pz = ctx->t_pz.as<float>();
    t.rgba = ctx->t_rgba.as<uint32_t>(); t.w = ctx->t_w.as<float>();
    t.flags = ctx->t_flags.as<uint8_t>(); t.bkey = ctx->t_bkey.as<uint64_t>();
  } else {
    t.px = ctx->u_px.as<float>(); t.py = ctx->u_py.as<float>(); t.pz = ctx->u_pz.as<float>();
    t.rgba = ctx->u_rgba.as<uint32_t>(); t.w = ctx->u_w.as<float>();
    t.flags = ctx->u_flags.as<uint8_t>(); t.bkey = ctx->u_bkey.as<uint64_t>();
  }
  t.R = R;
  return t;
}

int ensure_tab(vbx_ctx* ctx, bool second, size_t R, bool with_bkey) {
  const size_t n = R + 1;
  if (!second) {
    HIP_TRY(ctx->t_px.ensure(n * 4)); HIP_TRY(ctx->t_py.ensure(n * 4)); HIP_TRY(ctx->t_pz.ensure(n * 4));
    HIP_TRY(ctx->t_rgba.ensure(n * 4)); HIP_TRY(ctx->t_w.ensure(n * 4)); HIP_TRY(ctx->t_flags.ensure(n));
    if (with_bkey) HIP_TRY(ctx->t_bkey.ensure(n * 8));
  } else {
    HIP_TRY(ctx->u_px.ensure(n * 4)); HIP_TRY(ctx->u_py.ensure(n * 4)); HIP_TRY(ctx->u_pz.ensure(n * 4));
    HIP_TRY(ctx->u_rgba.ensure(n * 4)); HIP_TRY(ctx->u_w.ensure(n * 4)); HIP_TRY(ctx->u_flags.ensure(n));
    if (with_bkey) HIP_TRY(ctx->u_bkey.ensure(n * 8));
  }
  return VBX_OK;
}

// rocPRIM is used only for the two generic primitives of the pipeline (LSD radix sort,
// exclusive scan); everything domain-specific is a kernel in this file.
// Frame-sized inputs (3e5..1e6 keys) sit below rocPRIM's default merge-sort limit, where it runs
// ~20 launch-bound merge passes over the full 64-bit key; the callers here only need a STABLE
// sort on a 20..26-bit field (the inputs are already in visiting order), which is 3-4 onesweep
// passes.  MergeSortLimit = 0 selects the LSD onesweep path for every size.
using SortCfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                           rocprim::default_config, 0>;
int sort_keys(vbx_ctx* ctx, uint64_t* in, uint64_t* out, size_t n, unsigned begin_bit,
              unsigned end_bit) {
  size_t tmp = 0;
  HIP_TRY(rocprim::radix_sort_keys<SortCfg>(nullptr, tmp, in, out, n, begin_bit, end_bit, ctx->stream));
  HIP_TRY(ctx->b_tmp.ensure(tmp));
  HIP_TRY(rocprim::radix_sort_keys<SortCfg>(ctx->b_tmp.p, tmp, in, out, n, begin_bit, end_bit, ctx->stream));
  return VBX_OK;
}
int sort_pairs(vbx_ctx* ctx, uint64_t* kin, uint64_t* kout, uint32_t* vin, uint32_t* vout, size_t n,
               unsigned begin_bit, unsigned end_bit) {
  size_t tmp = 0;
  HIP_TRY(rocprim::radix_sort_pairs<SortCfg>(nullptr, tmp, kin, kout, vin, vout, n, begin_bit, end_bit,
                                             ctx->stream));
  HIP_TRY(ctx->b_tmp.ensure(tmp));
  HIP_TRY(rocprim::radix_sort_pairs<SortCfg>(ctx->b_tmp.p, tmp, kin, kout, vin, vout, n, begin_bit, end_bit,
                                             ctx->stream));
  return VBX_OK;
}
int exclusive_scan_u32(vbx_ctx* ctx, uint32_t* in, uint32_t* out, size_t n) {
  size_t tmp = 0;
  HIP_TRY(rocprim::exclusive_scan(nullptr, tmp, in, out, 0u, n, rocprim::plus<uint32_t>(), ctx->stream));
  HIP_TRY(ctx->b_tmp.ensure(tmp));
  HIP_TRY(rocprim::exclusive_scan(ctx->b_tmp.p, tmp, in, out, 0u, n, rocprim::plus<uint32_t>(),
                                  ctx->stream));
  return VBX_OK;
}

inline unsigned bits_for(uint64_t v) {
  unsigned b = 1;
  while (b < 64 && (v >> b)) ++b;
  return b;
}

void tmark(vbx_ctx* ctx, int i) {
  if (ctx->timing) {
    (void)hipEventRecord(ctx->ev[i], ctx->stream);
    ctx->ev_hit[i] = true;
  }
}

CastCfg make_cast_cfg(vbx_ctx* ctx, const vbx_tsdf_cfg* cfg, const float pos[3]) {
  CastCfg c{};
  c.origin = {pos[0], pos[1], pos[2]};
  c.trunc = cfg->default_truncation_distance;
  c.max_ray_length_m = cfg->max_ray_length_m;
  c.exp = getenv("VBX_EXP") ? atoi(getenv("VBX_EXP")) : 0;
  c.min_ray_length_m = cfg->min_ray_length_m;
  c.max_weight = cfg->max_weight;
  c.sparsity_factor = cfg->sparsity_compensation_factor;
  c.carving = cfg->voxel_carving_enabled != 0;
  // tsdf_integrator.cc:62-65: clearing rays have no utility if voxel_carving is disabled
  c.allow_clear = (cfg->allow_clear != 0) && (cfg->voxel_carving_enabled != 0);
  c.use_const_weight = cfg->use_const_weight != 0;
  c.dropoff = cfg->use_weight_dropoff != 0;
  c.sparsity = cfg->use_sparsity_compensation_factor != 0;
  c.anti_grazing = cfg->enable_anti_grazing != 0;
  c.max_consecutive = cfg->max_consecutive_ray_collisions;
  c.start_factor_times_inv = cfg->start_voxel_subsampling_factor * ctx->map.voxel_size_inv;
  return c;
}


// Visiting order of the points: nullptr = MixedThreadSafeIndex (closed form on the device), else
// a device array s_of_p for "sorted" (integration_order_mode, tsdf_integrator.h:72-74).
int visiting_order(vbx_ctx* ctx, const vbx_tsdf_cfg* cfg, const float* d_pts, size_t n, const uint32_t** order) {
  *order = nullptr;
  if (cfg->integration_order_mode == 0) return VBX_OK;
  HIP_TRY(ctx->b_keys0.ensure(n * 8));
  HIP_TRY(ctx->b_keys1.ensure(n * 8));
  HIP_TRY(ctx->b_order.ensure(n * 4));
  hipLaunchKernelGGL(k_sorted_keys, grid_for(n), dim3(256), 0, ctx->stream, d_pts, n, ctx->b_keys0.as<uint64_t>());
  int rc = sort_keys(ctx, ctx->b_keys0.as<uint64_t>(), ctx->b_keys1.as<uint64_t>(), n, 0, 64);
  if (rc) return rc;
  hipLaunchKernelGGL(k_sorted_inverse, grid_for(n), dim3(256), 0, ctx->stream, ctx->b_keys1.as<uint64_t>(), n,
                     ctx->b_order.as<uint32_t>());
  *order = ctx->b_order.as<uint32_t>();
  return VBX_OK;
}

// Order the emitted (voxel, order) keys and fold them per voxel (keys in b_keys0).
int sort_and_fold(vbx_ctx* ctx, const RayTab& tab, const CastCfg& c, uint32_t total) {
  MapDev& m = ctx->map;
  hipStream_t s = ctx->stream;
  // gids are below pool_used * nvox; h_state holds the value from after this call's slot
  // assignment (every path reads DevState back between k_commit_alloc and here)
  const unsigned end_bit = 32 + bits_for((uint64_t)std::max<uint32_t>(ctx->h_state.pool_used, 1) * m.nvox);
  // keys are emitted ray by ray in visiting order, so a stable sort on the voxel field alone
  // leaves every voxel's updates in visiting order (invalid keys, all ones, go last)
  int rc = sort_keys(ctx, ctx->b_keys0.as<uint64_t>(), ctx->b_keys1.as<uint64_t>(), total, 32,
                     std::min(64u, end_bit + 1));
  if (rc) return rc;
  tmark(ctx, 5);
  // long runs are collected by k_fold and folded wave-cooperatively afterwards (their number is
  // bounded by total / kFoldShort)
  HIP_TRY(ctx->b_long.ensure(((size_t)total / kFoldShort + 2) * 4));
  HIP_TRY(hipMemsetAsync(&ctx->d_state->fold_long_count, 0, 4, s));
  hipLaunchKernelGGL(k_fold, grid_for(total), dim3(256), 0, s, ctx->b_keys1.as<uint64_t>(),
                     (size_t)total, tab, c, m, ctx->b_long.as<uint32_t>(), ctx->d_state);
  {
    const unsigned waves = (unsigned)std::min<size_t>(8192, (size_t)total / kFoldShort + 1);
    hipLaunchKernelGGL(k_fold_long, dim3((waves + 3) / 4), dim3(256), 0, s, ctx->b_keys1.as<uint64_t>(),
                       (size_t)total, tab, c, m, ctx->b_long.as<uint32_t>(), ctx->d_state);
  }
  tmark(ctx, 6);
  ctx->counters.voxel_updates = total;
  return VBX_OK;
}

// Shared tail of all three integrators: allocate blocks along the rays, emit ordered voxel
// keys, sort, fold.  `limit` (optional) bounds the number of voxels each ray visits.
int march_and_fold(vbx_ctx* ctx, const RayTab& tab, const CastCfg& c, bool from_origin,
                   const uint32_t* limit, bool blocks_already_marked, const uint64_t* graze_keys,
                   uint32_t n_graze) {
  const uint32_t R = tab.R;
  if (R == 0) return VBX_OK;
  MapDev& m = ctx->map;
  hipStream_t s = ctx->stream;

  HIP_TRY(ctx->b_cnt.ensure((size_t)(R + 1) * 4));
  HIP_TRY(ctx->b_off.ensure((size_t)(R + 1) * 4));
  hipLaunchKernelGGL(k_ray_count, grid_for(R + 1), dim3(256), 0, s, tab, c, m, from_origin ? 1 : 0,
                     limit, ctx->b_cnt.as<uint32_t>());
  int rc = exclusive_scan_u32(ctx, ctx->b_cnt.as<uint32_t>(), ctx->b_off.as<uint32_t>(), R + 1);
  if (rc) return rc;
  if (!blocks_already_marked) {
    hipLaunchKernelGGL(k_ray_mark_blocks, grid_for(R), dim3(256), 0, s, tab, c, m,
                       from_origin ? 1 : 0, limit, ctx->b_newlist.as<uint32_t>(), ctx->d_state);
    hipLaunchKernelGGL(k_assign_slots, grid_for(m.cap_blocks), dim3(256), 0, s, m,
                       ctx->b_newlist.as<uint32_t>(), ctx->d_state);
    hipLaunchKernelGGL(k_commit_alloc, dim3(1), dim3(1), 0, s, m, ctx->d_state);
    tmark(ctx, 2);
  }
  // total number of keys = off[R]
  uint32_t total = 0;
  rc = sync_state(ctx, ctx->b_off.as<uint32_t>() + R, &total);
  if (rc) return rc;
  rc = check_state_error(ctx);
  if (rc) return rc;
  if (total == 0) return VBX_OK;

  HIP_TRY(ctx->b_keys0.ensure((size_t)total * 8));
  HIP_TRY(ctx->b_keys1.ensure((size_t)total * 8));
  hipLaunchKernelGGL(k_ray_emit, grid_for(R), dim3(256), 0, s, tab, c, m, from_origin ? 1 : 0, limit,
                     ctx->b_off.as<uint32_t>(), ctx->b_keys0.as<uint64_t>(), graze_keys, n_graze,
                     ctx->d_state);
  hipLaunchKernelGGL(k_count_cast, grid_for(R), dim3(256), 0, s, tab.flags, R, ctx->d_state);
  tmark(ctx, 4);
  return sort_and_fold(ctx, tab, c, total);
}

int integrate_simple(vbx_ctx* ctx, const vbx_tsdf_cfg* cfg, const Pose& T, const float* d_pts,
                     const uint32_t* d_rgba, size_t n, int freespace) {
  CastCfg c = make_cast_cfg(ctx, cfg, &T.t.x);
  const uint32_t* order = nullptr;
  {
    int orc_ = visiting_order(ctx, cfg, d_pts, n, &order);
    if (orc_) return orc_;
  }
  int rc = ensure_tab(ctx, false, n, false);
  if (rc) return rc;
  RayTab tab = make_tab(ctx, false, (uint32_t)n);
  tab.bkey = nullptr;
  hipLaunchKernelGGL(k_prep_points, grid_for(n), dim3(256), 0, ctx->stream, d_pts, d_rgba, n, T, c,
                     freespace, tab, (float*)nullptr, (float*)nullptr, (float*)nullptr, order);
  tmark(ctx, 1);
  return march_and_fold(ctx, tab, c, /*from_origin=*/true, nullptr, false, nullptr, 0);
}

// MergedTsdfIntegrator::integrateVoxels walks voxel_map / clear_map — std::unordered_map keyed by
// GlobalIndex with LongIndexHash (block_hash.h:54-64) — from begin() to end()
// (tsdf_integrator.cc:440-456).  That order is a property of libstdc++'s hashtable (bucket
// count growth, node splicing) given the hash values and the insertion sequence, so it is
// obtained the way the reference obtains it: the bundle keys are inserted into the same container,
// in bundleRays' insertion order (first point of each bundle, visiting order), on the host.
// perm[rank in ascending key order] = row in visiting order; non-clearing bundles first (:324-333).
struct HostL3Hash {
  size_t operator()(const l3& k) const { return (size_t)long_index_hash(k); }
};
struct HostL3Eq {
  bool operator()(const l3& a, const l3& b) const { return a.x == b.x && a.y == b.y && a.z == b.z; }
};
int merged_reference_order(vbx_ctx* ctx, size_t n, uint32_t nb, const uint32_t** perm_out) {
  hipStream_t s = ctx->stream;
  HIP_TRY(ctx->b_bkeys.ensure((size_t)nb * 8));
  HIP_TRY(ctx->b_bfirst.ensure((size_t)nb * 4));
  HIP_TRY(ctx->b_bperm.ensure((size_t)nb * 4));
  hipLaunchKernelGGL(k_merged_collect, grid_for(n), dim3(256), 0, s, ctx->b_keys1.as<uint64_t>(),
                     ctx->b_vals1.as<uint32_t>(), ctx->b_head.as<uint32_t>(), ctx->b_rank.as<uint32_t>(),
                     (uint32_t)n, ctx->b_bkeys.as<uint64_t>(), ctx->b_bfirst.as<uint32_t>());
  std::vector<uint64_t> keys(nb);
  std::vector<uint32_t> first(nb), perm(nb), idx(nb);
  HIP_TRY(hipMemcpyAsync(keys.data(), ctx->b_bkeys.p, (size_t)nb * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(first.data(), ctx->b_bfirst.p, (size_t)nb * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  // the keys are sorted with the clearing bit on top: [0, n1) normal bundles, [n1, nb) clearing
  uint32_t n1 = 0;
  while (n1 < nb && !(keys[n1] >> 63)) ++n1;
  // insertion order = ascending visiting position of each bundle's first point; positions are
  // unique and < n, so a direct-address pass orders them without a comparison sort
  std::vector<int32_t>& by_s = ctx->h_by_s;
  by_s.assign(n, -1);
  for (uint32_t b = 0; b < nb; ++b) by_s[first[b]] = (int32_t)b;
  uint32_t q1 = 0, q2 = n1;
  for (size_t sidx = 0; sidx < n; ++sidx) {
    const int32_t b = by_s[sidx];
    if (b < 0) continue;
    if ((uint32_t)b < n1) idx[q1++] = (uint32_t)b; else idx[q2++] = (uint32_t)b;
  }
  uint32_t row = 0;
  for (int pass = 0; pass < 2; ++pass) {
    const uint32_t lo = pass ? n1 : 0, hi = pass ? nb : n1;
    // node storage from a monotonic arena: the allocator has no influence on the iteration order
    std::pmr::monotonic_buffer_resource arena((size_t)(hi - lo) * 64 + 4096);
    std::pmr::unordered_map<l3, uint32_t, HostL3Hash, HostL3Eq> map(&arena);
    for (uint32_t q = lo; q < hi; ++q) {
      const uint64_t k = keys[idx[q]] & ~(1ull << 63);
      const l3 g{(long long)(k & 0x1FFFFFu) - (1ll << 20), (long long)((k >> 21) & 0x1FFFFFu) - (1ll << 20),
                 (long long)((k >> 42) & 0x1FFFFFu) - (1ll << 20)};
      map.emplace(g, idx[q]);
    }
    for (const auto& kv : map) perm[kv.second] = row++;
  }
  HIP_TRY(hipMemcpyAsync(ctx->b_bperm.p, perm.data(), (size_t)nb * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));  // perm is a local
  *perm_out = ctx->b_bperm.as<uint32_t>();
  return VBX_OK;
}

int integrate_merged(vbx_ctx* ctx, const vbx_tsdf_cfg* cfg, const Pose& T, const float* d_pts,
                     const uint32_t* d_rgba, size_t n, int freespace) {
  CastCfg c = make_cast_cfg(ctx, cfg, &T.t.x);
  const uint32_t* order = nullptr;
  {
    int orc_ = visiting_order(ctx, cfg, d_pts, n, &order);
    if (orc_) return orc_;
  }
  hipStream_t s = ctx->stream;
  int rc = ensure_tab(ctx, false, n, false);
  if (rc) return rc;
  HIP_TRY(ctx->b_pcx.ensure(n * 4)); HIP_TRY(ctx->b_pcy.ensure(n * 4)); HIP_TRY(ctx->b_pcz.ensure(n * 4));
  RayTab pt = make_tab(ctx, false, (uint32_t)n);
  pt.bkey = nullptr;
  hipLaunchKernelGGL(k_prep_points, grid_for(n), dim3(256), 0, s, d_pts, d_rgba, n, T, c, freespace,
                     pt, ctx->b_pcx.as<float>(), ctx->b_pcy.as<float>(), ctx->b_pcz.as<float>(), order);
  // bundleRays (tsdf_integrator.cc:340-371): group points by endpoint voxel.  A stable sort
  // of (key, s) keeps each bundle's points in visiting order.
  HIP_TRY(ctx->b_keys0.ensure(n * 8)); HIP_TRY(ctx->b_keys1.ensure(n * 8));
  HIP_TRY(ctx->b_vals0.ensure(n * 4)); HIP_TRY(ctx->b_vals1.ensure(n * 4));
  hipLaunchKernelGGL(k_merged_keys, grid_for(n), dim3(256), 0, s, pt, (uint32_t)n, ctx->map,
                     ctx->b_keys0.as<uint64_t>(), ctx->b_vals0.as<uint32_t>());
  rc = sort_pairs(ctx, ctx->b_keys0.as<uint64_t>(), ctx->b_keys1.as<uint64_t>(),
                  ctx->b_vals0.as<uint32_t>(), ctx->b_vals1.as<uint32_t>(), n, 0, 64);
  if (rc) return rc;
  HIP_TRY(ctx->b_head.ensure((n + 1) * 4)); HIP_TRY(ctx->b_rank.ensure((n + 1) * 4));
  hipLaunchKernelGGL(k_merged_heads, grid_for(n + 1), dim3(256), 0, s, ctx->b_keys1.as<uint64_t>(),
                     (uint32_t)n, ctx->b_head.as<uint32_t>());
  rc = exclusive_scan_u32(ctx, ctx->b_head.as<uint32_t>(), ctx->b_rank.as<uint32_t>(), n + 1);
  if (rc) return rc;
  uint32_t nb = 0;
  rc = sync_state(ctx, ctx->b_rank.as<uint32_t>() + n, &nb);
  if (rc) return rc;
  if (nb == 0) return VBX_OK;
  rc = ensure_tab(ctx, true, nb, true);
  if (rc) return rc;
  HIP_TRY(ctx->b_graze.ensure((size_t)nb * 8));
  HIP_TRY(hipMemsetAsync(ctx->b_graze.p, 0xFF, (size_t)nb * 8, s));
  RayTab bt = make_tab(ctx, true, nb);
  const uint32_t* perm = nullptr;
  if (cfg->merged_bundle_order == 0) {
    rc = merged_reference_order(ctx, n, nb, &perm);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_merged_bundle, grid_for(n), dim3(256), 0, s, ctx->b_keys1.as<uint64_t>(),
                     ctx->b_vals1.as<uint32_t>(), ctx->b_head.as<uint32_t>(),
                     ctx->b_rank.as<uint32_t>(), (uint32_t)n, pt, ctx->b_pcx.as<float>(),
                     ctx->b_pcy.as<float>(), ctx->b_pcz.as<float>(), T, bt,
                     ctx->b_graze.as<uint64_t>(), perm, ctx->d_state);
  tmark(ctx, 1);
  // Non-clearing bundles sort before clearing ones (bit 63), so the graze key list is the
  // sorted prefix of non-clearing bundle keys; entries of clearing bundles stay ~0 (sorted last).
  const uint64_t* graze = c.anti_grazing ? ctx->b_graze.as<uint64_t>() : nullptr;
  return march_and_fold(ctx, bt, c, /*from_origin=*/true, nullptr, false, graze, nb);
}

int integrate_fast(vbx_ctx* ctx, const vbx_tsdf_cfg* cfg, const Pose& T, const float* d_pts,
                   const uint32_t* d_rgba, size_t n, int freespace) {
  CastCfg c = make_cast_cfg(ctx, cfg, &T.t.x);
  const uint32_t* order = nullptr;
  {
    int orc_ = visiting_order(ctx, cfg, d_pts, n, &order);
    if (orc_) return orc_;
  }
  hipStream_t s = ctx->stream;
  MapDev& m = ctx->map;
  constexpr uint32_t kSetSize = (1u << 20) + 10000u;
  if (!ctx->startset_init) {
    HIP_TRY(ctx->b_startset.ensure((size_t)kSetSize * 4));
    HIP_TRY(hipMemsetAsync(ctx->b_startset.p, 0, (size_t)kSetSize * 4, s));
    ctx->startset_init = true;
    ctx->start_offset = 0;
    ctx->start_sentinel_live = true;
  }
  // tsdf_integrator.cc:564-569 + ApproxHashSet::resetApproxSet (approx_hash_array.h:156-169)
  if ((++ctx->reset_counter) >= cfg->clear_checks_every_n_frames) {
    ctx->reset_counter = 0;
    ++ctx->obs_epoch;  // voxel_observed set cleared (exact-set form of resetApproxSet)
    if (++ctx->obsset_offset >= 10000u) {  // both sets reset together (tsdf_integrator.cc:566-568)
      if (ctx->obsset_init) HIP_TRY(hipMemsetAsync(ctx->b_obsset.p, 0, (size_t)kSetSize * 4, s));
      ctx->obsset_offset = 0;
      ctx->obsset_sentinel_live = true;
    }
    if (++ctx->start_offset >= 10000u) {
      HIP_TRY(hipMemsetAsync(ctx->b_startset.p, 0, (size_t)kSetSize * 4, s));
      ctx->start_offset = 0;
      ctx->start_sentinel_live = true;
    }
  }

  int rc = ensure_tab(ctx, false, n, false);
  if (rc) return rc;
  RayTab pt = make_tab(ctx, false, (uint32_t)n);
  pt.bkey = nullptr;
  hipLaunchKernelGGL(k_prep_points, grid_for(n), dim3(256), 0, s, d_pts, d_rgba, n, T, c, freespace,
                     pt, (float*)nullptr, (float*)nullptr, (float*)nullptr, order);
  HIP_TRY(ctx->b_keys0.ensure(n * 8)); HIP_TRY(ctx->b_keys1.ensure(n * 8));
  HIP_TRY(ctx->b_vals0.ensure(n * 4)); HIP_TRY(ctx->b_vals1.ensure(n * 4));
  hipLaunchKernelGGL(k_fast_keys, grid_for(n), dim3(256), 0, s, pt, (uint32_t)n, c,
                     ctx->b_keys0.as<uint64_t>(), ctx->b_vals0.as<uint32_t>());
  // keys[s] = slot << 32 | s is written in visiting order: a stable sort on the 20 slot bits
  // (+ bit 52, set only in the all-ones key of dropped points) orders by (slot, s)
  rc = sort_pairs(ctx, ctx->b_keys0.as<uint64_t>(), ctx->b_keys1.as<uint64_t>(),
                  ctx->b_vals0.as<uint32_t>(), ctx->b_vals1.as<uint32_t>(), n, 32, 53);
  if (rc) return rc;
  hipLaunchKernelGGL(k_fast_start_dedupe, grid_for(n), dim3(256), 0, s, ctx->b_keys1.as<uint64_t>(),
                     ctx->b_vals1.as<uint32_t>(), (uint32_t)n, ctx->b_startset.as<uint32_t>(),
                     ctx->start_offset, ctx->start_sentinel_live ? 1 : 0, pt.flags);
  hipLaunchKernelGGL(k_fast_start_commit, grid_for(n), dim3(256), 0, s, ctx->b_keys1.as<uint64_t>(),
                     ctx->b_vals1.as<uint32_t>(), (uint32_t)n, ctx->b_startset.as<uint32_t>(),
                     ctx->start_offset, ctx->d_state);
  HIP_TRY(ctx->b_head.ensure((n + 1) * 4)); HIP_TRY(ctx->b_rank.ensure((n + 1) * 4));
  hipLaunchKernelGGL(k_compact_flags, grid_for(n + 1), dim3(256), 0, s, pt.flags, (uint32_t)n,
                     ctx->b_head.as<uint32_t>());
  rc = exclusive_scan_u32(ctx, ctx->b_head.as<uint32_t>(), ctx->b_rank.as<uint32_t>(), n + 1);
  if (rc) return rc;
  rc = ensure_tab(ctx, true, n, false);
  if (rc) return rc;
  RayTab kt = make_tab(ctx, true, 0);
  kt.bkey = nullptr;
  hipLaunchKernelGGL(k_compact_rays, grid_for(n), dim3(256), 0, s, pt, ctx->b_head.as<uint32_t>(),
                     ctx->b_rank.as<uint32_t>(), (uint32_t)n, kt, ctx->d_state);
  rc = sync_state(ctx);
  if (rc) return rc;
  if (ctx->h_state.sentinel_cleared) ctx->start_sentinel_live = false;
  const uint32_t R = (uint32_t)ctx->h_state.num_kept;
  kt.R = R;
  tmark(ctx, 1);
  if (R == 0) return VBX_OK;

  // Candidate blocks along the full (unterminated) paths; a block only becomes part of the
  // Layer ("published") when a ray actually reaches it (k_fast_emit).
  HIP_TRY(ctx->b_cnt.ensure((size_t)(R + 1) * 4));
  HIP_TRY(ctx->b_off.ensure((size_t)(R + 1) * 4));
  hipLaunchKernelGGL(k_ray_count, grid_for(R + 1), dim3(256), 0, s, kt, c, m, 0,
                     (const uint32_t*)nullptr, ctx->b_cnt.as<uint32_t>());
  rc = exclusive_scan_u32(ctx, ctx->b_cnt.as<uint32_t>(), ctx->b_off.as<uint32_t>(), R + 1);
  if (rc) return rc;
  // Every ray emits at most sqrt(3) * (max_ray_length + truncation) / voxel_size + 4 voxels
  // (L1 <= sqrt(3) L2 of the walked segment), so the list buffer is sized without waiting for
  // the exact total; 288 GB of HBM make the slack irrelevant and the buffer is reused.
  const double seg = (double)c.max_ray_length_m + (double)c.trunc;
  const size_t per_ray = (size_t)(1.7320508075688772 * seg * (double)m.voxel_size_inv) + 6;
  const size_t vox_cap = std::min<size_t>((size_t)R * per_ray + 64, 0xFFFFFFF0u);
  HIP_TRY(ctx->b_vox.ensure(vox_cap * 4));
  HIP_TRY(ctx->b_redo.ensure((size_t)(R + 1) * 4));
  hipLaunchKernelGGL(k_fast_build_lists<kListRPW>, dim3((R + 4 * kListRPW - 1) / (4 * kListRPW)), dim3(256), 0, s, kt, c, m, ctx->b_off.as<uint32_t>(),
                     ctx->b_vox.as<uint32_t>(), (uint32_t)vox_cap, ctx->b_newlist.as<uint32_t>(),
                     (const uint32_t*)nullptr, ctx->b_redo.as<uint32_t>(), ctx->d_state);
  hipLaunchKernelGGL(k_assign_slots, grid_for(m.cap_blocks), dim3(256), 0, s, m,
                     ctx->b_newlist.as<uint32_t>(), ctx->d_state);
  hipLaunchKernelGGL(k_commit_alloc, dim3(1), dim3(1), 0, s, m, ctx->d_state);
  // second pass over the queued rays (grid sized for the first-frame worst case; idle
  // workgroups leave at once); capacity / lookup errors surface at the solver's first check
  if (ctx->fast_redo_grid == 0) ctx->fast_redo_grid = R;  // first frame: every block is new
  hipLaunchKernelGGL(k_fast_build_lists<kListRPW>, dim3((std::max<uint32_t>(ctx->fast_redo_grid, 1024) + 4 * kListRPW - 1) / (4 * kListRPW)), dim3(256), 0, s, kt, c, m,
                     ctx->b_off.as<uint32_t>(), ctx->b_vox.as<uint32_t>(), (uint32_t)vox_cap,
                     ctx->b_newlist.as<uint32_t>(), ctx->b_redo.as<uint32_t>(), (uint32_t*)nullptr, ctx->d_state);
  tmark(ctx, 2);

  // claim arrays + tags (see k_fast_sweep)
  const size_t nvox_total = (size_t)m.cap_blocks * m.nvox;
  // ray-index bits: sized by the cloud (an upper bound of R) so the tag layout — and with it
  // the claim arrays' contents — stays valid from frame to frame
  const int s_bits = std::max((int)bits_for(std::max<size_t>(n, 2) - 1), ctx->own_s_bits);
  const bool fresh = (ctx->b_own0.p == nullptr);
  HIP_TRY(ctx->b_own0.ensure(nvox_total * 4));
  HIP_TRY(ctx->b_own1.ensure(nvox_total * 4));
  HIP_TRY(ctx->b_cl.ensure(nvox_total * 4));
  const uint32_t max_tag = (1u << (32 - s_bits)) - 2;
  auto reset_tags = [&]() -> int {
    HIP_TRY(hipMemsetAsync(ctx->b_own0.p, 0xFF, nvox_total * 4, s));
    HIP_TRY(hipMemsetAsync(ctx->b_own1.p, 0xFF, nvox_total * 4, s));
    HIP_TRY(hipMemsetAsync(ctx->b_cl.p, 0xFF, nvox_total * 4, s));
    ctx->own_s_bits = s_bits;
    ctx->own_tag = max_tag;
    return VBX_OK;
  };
  if (fresh || s_bits != ctx->own_s_bits || ctx->own_tag < 1024) {
    rc = reset_tags();
    if (rc) return rc;
  }
  const bool strict_set = cfg->fast_observed_set == 0;  // the reference's ApproxHashSet semantics
  const bool keep_observed = cfg->clear_checks_every_n_frames > 1 && !strict_set;
  if (keep_observed && ctx->b_obs.cap < nvox_total * 4) {
    HIP_TRY(ctx->b_obs.ensure(nvox_total * 4));
    HIP_TRY(hipMemsetAsync(ctx->b_obs.p, 0, nvox_total * 4, s));
  }
  HIP_TRY(ctx->b_T.ensure((size_t)(R + 1) * 4));
  HIP_TRY(ctx->b_TH.ensure((size_t)(R + 1) * 4));
  HIP_TRY(ctx->b_U.ensure((size_t)(R + 1) * 4));
  HIP_TRY(ctx->b_rank.ensure((size_t)(R + 1) * 4));
  HIP_TRY(ctx->b_act0.ensure((size_t)(R + 1) * 4));
  HIP_TRY(ctx->b_act1.ensure((size_t)(R + 1) * 4));
  uint32_t iters_total = 0;
  auto run_solver = [&]() -> int {
    SweepArgs sa{};
    sa.off = ctx->b_off.as<uint32_t>();
    sa.vox = ctx->b_vox.as<uint32_t>();
    sa.cl = ctx->b_cl.as<uint32_t>();
    sa.tag_cl = --ctx->own_tag;
    sa.s_bits = s_bits;
    sa.max_consecutive = c.max_consecutive;
    sa.TL = ctx->b_T.as<uint32_t>();
    sa.TH = ctx->b_TH.as<uint32_t>();
    sa.U = ctx->b_U.as<uint32_t>();
    sa.obs = keep_observed ? ctx->b_obs.as<uint32_t>() : nullptr;
    sa.obs_epoch = ctx->obs_epoch;
    // sweep 0: publish the full-path possible claims (TH = path length);
    // sweep 1: lower bounds only (TH cannot move while there are no certain claims);
    // sweeps 2..: both bounds, open rays only.  (A persistent tail kernel with grid barriers
    // instead of launches was measured slower: 1.07 vs 0.98 ms — barrier + L2 write-back per
    // sweep cost more than a launch.)
    uint32_t iters = 0;
    uint32_t n_open = R;  // host-side upper bound of the open list
    uint32_t tag_rd = 0xFFFFFFFFu;
    int ch_flip = 0;      // which of the two possible-claim arrays holds the readable sweep
    int list_sel = 0;     // which work list the next sweep reads
    int cnt_cur = 0;      // act_count index of that list's length
    bool have_list = false;
    uint32_t* lists[2] = {ctx->b_act0.as<uint32_t>(), ctx->b_act1.as<uint32_t>()};
    uint32_t* chs[2] = {ctx->b_own0.as<uint32_t>(), ctx->b_own1.as<uint32_t>()};
    for (;;) {
      {
        // sweeps per host check (an idle sweep is ~5 us, a check ~30 us): the first three give
        // the open-ray count that sizes the later grids; then one batch up to where the previous
        // frame converged (consecutive frames behave alike), then fours
        int kBatch = (iters == 0) ? 3 : 4;
        if (iters == 3 && ctx->fast_last_iters > 7) kBatch = (int)ctx->fast_last_iters - 3 + 1;
        for (int b = 0; b < kBatch; ++b) {
          sa.init = (iters == 0) ? 1 : 0;
          sa.sweep_idx = iters;
          sa.l_only = (iters == 1) ? 1 : 0;
          const bool writes_ch = !sa.l_only;
          const bool writes_list = iters >= 2;
          sa.cnt_in = cnt_cur;
          sa.cnt_out = (cnt_cur + 1) % 3;
          sa.list_in = have_list ? lists[list_sel] : nullptr;
          sa.list_out = writes_list ? lists[have_list ? (list_sel ^ 1) : 0] : nullptr;
          sa.n_in = n_open;
          sa.ch_rd = chs[ch_flip];
          sa.ch_wr = chs[ch_flip ^ 1];
          sa.tag_rd = tag_rd;
          sa.tag_wr = writes_ch ? --ctx->own_tag : 0;
          if (writes_list && !have_list) HIP_TRY(hipMemsetAsync(&ctx->d_state->act_count[sa.cnt_out], 0, 4, s));
          if (iters == 0)
            hipLaunchKernelGGL(k_fast_sweep<64>, grid_for((size_t)n_open * 64), dim3(256), 0, s, sa, R, ctx->d_state);
          else if (n_open <= 8192)  // few open rays: a whole wave per ray (64 list entries per step)
            hipLaunchKernelGGL(k_fast_sweep<64>, grid_for((size_t)n_open * 64), dim3(256), 0, s, sa, R, ctx->d_state);
          else
            hipLaunchKernelGGL(k_fast_sweep<16>, grid_for((size_t)n_open * 16), dim3(256), 0, s, sa, R, ctx->d_state);
          if (writes_ch) {
            tag_rd = sa.tag_wr;
            ch_flip ^= 1;
          }
          if (writes_list) {
            list_sel = have_list ? (list_sel ^ 1) : 0;
            have_list = true;
            cnt_cur = sa.cnt_out;
          }
          ++iters;
        }
        rc = sync_state(ctx);
        if (rc) return rc;
        rc = check_state_error(ctx);
        if (rc) return rc;
        n_open = ctx->h_state.act_count[cnt_cur];
        // size the next frame's second list-building pass (it is a grid-stride loop, so this is
        // only a performance hint)
        ctx->fast_redo_grid = std::max<uint32_t>(1, 2 * ctx->h_state.redo_count);
      }
      if (getenv("VBX_DEBUG")) fprintf(stderr, "[vbx] fast solver: after %u sweeps %u open rays of %u\n", iters, n_open, R);
      if (n_open == 0) break;
      if (iters > 1000000 || ctx->own_tag < 128) {
        ctx->fail("Fast integrator: early-termination solver did not converge");
        return VBX_ERR_HIP;
      }
    }
    // sweeps that did work (the launches after convergence are idle)
    if (ctx->h_state.fast_idle_sweep) iters = std::min(iters, 0xFFFFFFFFu - ctx->h_state.fast_idle_sweep);
    iters_total = iters;
    ctx->fast_last_iters = std::min<uint32_t>(iters, 64);
    return VBX_OK;
  };
  rc = run_solver();
  if (rc) return rc;
  if (keep_observed)
    hipLaunchKernelGGL(k_fast_mark_observed, grid_for((size_t)R * 16), dim3(256), 0, s, ctx->b_off.as<uint32_t>(),
                       ctx->b_vox.as<uint32_t>(), ctx->b_T.as<uint32_t>(), R, ctx->b_obs.as<uint32_t>(),
                       ctx->obs_epoch);
  if (strict_set) {
    tmark(ctx, 8);
    // refinement rounds (see k_strict_keys): T lives in b_T / b_TH alternately, probe offsets in b_cnt
    if (!ctx->obsset_init) {
      HIP_TRY(ctx->b_obsset.ensure((size_t)kSetSize * 4));
      HIP_TRY(hipMemsetAsync(ctx->b_obsset.p, 0, (size_t)kSetSize * 4, s));
      ctx->obsset_init = true;
    }
    uint32_t* Tcur = ctx->b_T.as<uint32_t>();
    uint32_t* Tnext = ctx->b_TH.as<uint32_t>();
    uint32_t* poff = ctx->b_cnt.as<uint32_t>();
    HIP_TRY(hipMemsetAsync(Tcur + R, 0, 4, s));
    uint32_t rounds = 0;
    uint32_t P = 0;
    for (;;) {
      rc = exclusive_scan_u32(ctx, Tcur, poff, R + 1);
      if (rc) return rc;
      rc = sync_state(ctx, poff + R, &P);  // also carries `changed` of the previous round
      if (rc) return rc;
      if (getenv("VBX_DEBUG") && rounds > 0)
        fprintf(stderr, "[vbx] strict round %u: %u probes, %u rays moved (%u grew)\n", rounds, P, ctx->h_state.act_count[0],
                ctx->h_state.act_count[1]);
      if (rounds > 0 && !ctx->h_state.changed) break;
      if (rounds > 4096) {
        ctx->fail("Fast integrator: observed-set replay did not converge");
        return VBX_ERR_HIP;
      }
      HIP_TRY(ctx->b_keys0.ensure((size_t)std::max<uint32_t>(P, 1) * 8));
      HIP_TRY(ctx->b_keys1.ensure((size_t)std::max<uint32_t>(P, 1) * 8));
      HIP_TRY(ctx->b_collided.ensure((size_t)std::max<uint32_t>(P, 1)));
      if (P) {
        hipLaunchKernelGGL(k_strict_keys, grid_for(P), dim3(256), 0, s, poff, R, P, ctx->b_off.as<uint32_t>(),
                           ctx->b_vox.as<uint32_t>(), m, ctx->b_keys0.as<uint64_t>());
        rc = sort_keys(ctx, ctx->b_keys0.as<uint64_t>(), ctx->b_keys1.as<uint64_t>(), P, 44, 64);
        if (rc) return rc;
        hipLaunchKernelGGL(k_strict_outcome, grid_for(P), dim3(256), 0, s, ctx->b_keys1.as<uint64_t>(), P,
                           ctx->b_obsset.as<uint32_t>(), ctx->obsset_offset, ctx->obsset_sentinel_live ? 1 : 0,
                           ctx->b_collided.as<uint8_t>());
      }
      HIP_TRY(hipMemsetAsync(&ctx->d_state->changed, 0, 4, s));
      HIP_TRY(hipMemsetAsync(&ctx->d_state->act_count[0], 0, 8, s));
      hipLaunchKernelGGL(k_strict_scan, grid_for((size_t)(R + 1) * 16), dim3(256), 0, s, poff, ctx->b_off.as<uint32_t>(), R,
                         ctx->b_collided.as<uint8_t>(), c.max_consecutive, Tcur, Tnext, ctx->b_U.as<uint32_t>(),
                         getenv("VBX_DEBUG") ? 1 : 0, ctx->d_state);
      std::swap(Tcur, Tnext);
      ++rounds;
    }
    // the sorted probe list of the last round (whose T equals the final T) is still in keys1
    if (P) {
      HIP_TRY(hipMemsetAsync(&ctx->d_state->sentinel_cleared, 0, 4, s));
      hipLaunchKernelGGL(k_strict_commit, grid_for(P), dim3(256), 0, s, ctx->b_keys1.as<uint64_t>(), P,
                         ctx->b_obsset.as<uint32_t>(), ctx->obsset_offset, ctx->d_state);
    }
    ctx->counters.replay_rounds = rounds;
  }
  uint32_t total = 0;
  // offsets of the keys each ray emits
  rc = exclusive_scan_u32(ctx, ctx->b_U.as<uint32_t>(), ctx->b_rank.as<uint32_t>(), R + 1);
  if (rc) return rc;
  rc = sync_state(ctx, ctx->b_rank.as<uint32_t>() + R, &total);
  if (rc) return rc;
  if (strict_set && ctx->h_state.sentinel_cleared) ctx->obsset_sentinel_live = false;
  ctx->counters.iterations = iters_total;
  tmark(ctx, 3);
  hipLaunchKernelGGL(k_count_cast, grid_for(R), dim3(256), 0, s, kt.flags, R, ctx->d_state);
  if (total == 0) return VBX_OK;
  HIP_TRY(ctx->b_keys0.ensure((size_t)total * 8));
  HIP_TRY(ctx->b_keys1.ensure((size_t)total * 8));
  hipLaunchKernelGGL(k_fast_emit, grid_for(total), dim3(256), 0, s, ctx->b_off.as<uint32_t>(),
                     ctx->b_vox.as<uint32_t>(), ctx->b_rank.as<uint32_t>(), R, total, m,
                     ctx->b_keys0.as<uint64_t>(), ctx->d_state);
  tmark(ctx, 4);
  return sort_and_fold(ctx, kt, c, total);
}

int integrate_device(vbx_ctx* ctx, int kind, const vbx_tsdf_cfg* cfg, const float pos[3],
                     const float quat[4], const float* d_pts, const uint8_t* d_rgba, size_t n,
                     int freespace) {
  if (!cfg || !pos || !quat || (n && (!d_pts || !d_rgba))) {
    ctx->fail("vbx_tsdf_integrate: null argument");
    return VBX_ERR_INVALID;
  }
  if (n >= (1ull << 31)) {
    ctx->fail("vbx_tsdf_integrate: too many points");
    return VBX_ERR_INVALID;
  }
  if (cfg->integration_order_mode != 0 && cfg->integration_order_mode != 1) {
    ctx->fail("Unknown integration order mode");  // integrator_utils.cc:12
    return VBX_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  ctx->counters = vbx_counters{};
  ctx->counters.points = n;
  if (n == 0) return VBX_OK;
  // per-call device counters
  hipLaunchKernelGGL(k_reset_call_state, dim3(1), dim3(1), 0, ctx->stream, ctx->d_state);
  Pose T;
  T.t = {pos[0], pos[1], pos[2]};
  T.qw = quat[0]; T.qx = quat[1]; T.qy = quat[2]; T.qz = quat[3];
  for (int i = 0; i < 9; ++i) ctx->ev_hit[i] = false;
  tmark(ctx, 0);
  int rc;
  const uint32_t* rgba32 = reinterpret_cast<const uint32_t*>(d_rgba);
  switch (kind) {
    case VBX_TSDF_SIMPLE: rc = integrate_simple(ctx, cfg, T, d_pts, rgba32, n, freespace); break;
    case VBX_TSDF_MERGED: rc = integrate_merged(ctx, cfg, T, d_pts, rgba32, n, freespace); break;
    case VBX_TSDF_FAST: rc = integrate_fast(ctx, cfg, T, d_pts, rgba32, n, freespace); break;
    default:
      ctx->fail("unknown TSDF integrator type %d", kind);  // tsdf_integrator.cc:40-43
      return VBX_ERR_INVALID;
  }
  if (rc) return rc;
  tmark(ctx, 7);
  rc = sync_state(ctx);
  if (rc) return rc;
  rc = check_state_error(ctx);
  if (rc) return rc;
  ctx->counters.rays_cast = ctx->h_state.rays_cast;
  ctx->counters.voxels_touched = ctx->h_state.voxels_touched;
  ctx->counters.blocks_allocated = ctx->h_state.blocks_published;
  if (ctx->timing) {
    (void)hipEventSynchronize(ctx->ev[7]);  // the state read-back spins on mapped memory; the runtime may not have retired the events yet
    float t[8] = {0};
    int last = 0;
    for (int i = 1; i < 8; ++i) {  // a stage a path skips reads as zero-length
      if (!ctx->ev_hit[i]) continue;
      (void)hipEventElapsedTime(&t[i], ctx->ev[last], ctx->ev[i]);
      last = i;
    }
    vbx_timing& o = ctx->last_timing;
    o.prep_ms = t[1]; o.alloc_ms = t[2]; o.solve_ms = t[3]; o.emit_ms = t[4];
    o.sort_ms = t[5]; o.fold_ms = t[6];
    o.replay_ms = 0.0f;
    if (ctx->ev_hit[8] && ctx->ev_hit[3]) {  // stage 3 = exact-set solve + reference-set replay rounds
      (void)hipEventElapsedTime(&o.replay_ms, ctx->ev[8], ctx->ev[3]);
      o.solve_ms = t[3] - o.replay_ms;
    }
    (void)hipEventElapsedTime(&o.total_ms, ctx->ev[0], ctx->ev[7]);
  }
  return VBX_OK;
}


__global__ void k_esdf_reset_flags(MapDev m, EsdfDev e, uint32_t n_slots, int drop_layer, int keep_classify_pending) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  // blocks addNewRobotPosition left work in: 8 = take part in this update (their voxels sit in
  // open_/raise_), 16 = also re-run the TSDF classification on them
  const uint32_t f = m.blk_flags[s];
  const uint32_t pend = f & (kFlagEsdfPendClassify | kFlagEsdfPendOpen);
  e.active[s] = (pend && !drop_layer) ? (8u | ((pend & kFlagEsdfPendClassify) ? 16u : 0u)) : 0u;
  // updateFromTsdfBlocks does not consume updated_blocks_: the classification stays pending
  uint32_t nf = f & ~((keep_classify_pending ? 0u : kFlagEsdfPendClassify) | kFlagEsdfPendOpen);
  if (drop_layer) nf &= ~(kFlagEsdfAlloc | (kFlagUpdMask << kFlagEsdfUpdShift));
  if (nf != f) m.blk_flags[s] = nf;
}
__global__ void k_esdf_clear_tsdf_bit(MapDev m, EsdfDev e, uint32_t n_slots) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  if (e.active[s] & 8u) m.blk_flags[s] &= ~4u;  // updated().reset(Update::kEsdf), esdf_integrator.cc:113-121
}

EsdfDev esdf_dev(vbx_ctx* ctx) {
  EsdfDev e;
  e.dist = ctx->b_edist.as<float>();
  e.state = ctx->b_estate.as<uint32_t>();
  e.raised = ctx->b_eraised.as<uint8_t>();
  e.active = ctx->b_eactive.as<uint32_t>();
  return e;
}

int esdf_ensure(vbx_ctx* ctx) {
  if (ctx->esdf_init) return VBX_OK;
  const MapDev& m = ctx->map;
  const size_t nv = (size_t)m.cap_blocks * m.nvox;
  HIP_TRY(ctx->b_edist.ensure(nv * 4));
  HIP_TRY(ctx->b_estate.ensure(nv * 4));
  HIP_TRY(ctx->b_eraised.ensure(nv));
  HIP_TRY(ctx->b_eactive.ensure((size_t)m.cap_blocks * 4));
  HIP_TRY(hipMemsetAsync(ctx->b_edist.p, 0, nv * 4, ctx->stream));
  HIP_TRY(hipMemsetAsync(ctx->b_estate.p, 0, nv * 4, ctx->stream));
  HIP_TRY(hipMemsetAsync(ctx->b_eraised.p, 0, nv, ctx->stream));
  HIP_TRY(hipMemsetAsync(ctx->b_eactive.p, 0, (size_t)m.cap_blocks * 4, ctx->stream));
  ctx->esdf_init = true;
  return VBX_OK;
}

template <int VPS>
int esdf_phase(vbx_ctx* ctx, const EsdfDev& e, const EsdfCfgDev& c, int mode, uint32_t used,
               uint32_t* sweeps) {
  hipStream_t s = ctx->stream;
  for (;;) {
    HIP_TRY(hipMemsetAsync(&ctx->d_state->changed, 0, 4, s));
    hipLaunchKernelGGL(k_esdf_tile<VPS>, dim3(used), dim3(kEsdfThreads), 0, s, ctx->map, e, c, mode, ctx->d_state);
    ++*sweeps;
    if (mode == 2) return VBX_OK;
    hipLaunchKernelGGL(k_esdf_rotate_active, grid_for(used), dim3(256), 0, s, e, used, 0);
    int rc = sync_state(ctx);
    if (rc) return rc;
    if (!ctx->h_state.changed) return VBX_OK;
    if (*sweeps > 100000) {
      ctx->fail("ESDF: wavefront did not converge");
      return VBX_ERR_HIP;
    }
  }
}

template <int VPS>
int esdf_update_t(vbx_ctx* ctx, const vbx_esdf_cfg* cfg, int batch, int clear_updated_flag,
                  const int32_t* list = nullptr, size_t n_list = 0, int list_incremental = 0) {
  MapDev& m = ctx->map;
  hipStream_t s = ctx->stream;
  int rc = esdf_ensure(ctx);
  if (rc) return rc;
  rc = sync_state(ctx);
  if (rc) return rc;
  const uint32_t used = ctx->h_state.pool_used;
  ctx->counters = vbx_counters{};
  if (used == 0) return VBX_OK;
  EsdfDev e = esdf_dev(ctx);
  EsdfCfgDev c;
  c.max_distance = cfg->max_distance_m;
  c.min_distance = cfg->min_distance_m;
  c.default_distance = cfg->default_distance_m;
  c.min_diff = cfg->min_diff_m;
  c.min_weight = cfg->min_weight;
  c.add_occupied_crust = cfg->add_occupied_crust != 0;
  c.voxel_size = m.voxel_size;
  const size_t nv = (size_t)used * m.nvox;
  for (int i = 0; i < 9; ++i) ctx->ev_hit[i] = false;
  tmark(ctx, 0);
  if (batch) {  // esdf_layer_->removeAllBlocks(), esdf_integrator.cc:95
    HIP_TRY(hipMemsetAsync(e.dist, 0, nv * 4, s));
    HIP_TRY(hipMemsetAsync(e.state, 0, nv * 4, s));
  }
  // `raised` is clear between updates except for the marks addNewRobotPosition left (a batch
  // update drops those with the layer)
  if (batch) HIP_TRY(hipMemsetAsync(e.raised, 0, nv, s));
  const bool robot_pending = ctx->esdf_robot_pending && !batch;
  ctx->esdf_robot_pending = false;
  hipLaunchKernelGGL(k_esdf_reset_flags, grid_for(used), dim3(256), 0, s, m, e, used, batch ? 1 : 0, list ? 1 : 0);
  HIP_TRY(hipMemsetAsync(&ctx->d_state->esdf_blocks, 0, 12, s));
  if (list) {
    HIP_TRY(ctx->b_head.ensure(std::max<size_t>(n_list, 1) * 12));
    HIP_TRY(hipMemcpyAsync(ctx->b_head.p, list, n_list * 12, hipMemcpyHostToDevice, s));
    if (n_list)
      hipLaunchKernelGGL(k_esdf_mark_listed, grid_for(n_list), dim3(256), 0, s, m, e, ctx->b_head.as<int32_t>(),
                         (uint32_t)n_list);
    hipLaunchKernelGGL(k_esdf_classify, dim3(used, (m.nvox + 255) / 256), dim3(256), 0, s, m, e, c,
                       list_incremental ? 1 : 0, 2, ctx->d_state);
  } else {
    hipLaunchKernelGGL(k_esdf_classify, dim3(used, (m.nvox + 255) / 256), dim3(256), 0, s, m, e, c,
                       batch ? 0 : 1, batch ? 0 : 1, ctx->d_state);
  }
  hipLaunchKernelGGL(k_esdf_seed_active, grid_for((size_t)used * 27), dim3(256), 0, s, m, e, used);
  rc = sync_state(ctx);
  if (rc) return rc;
  tmark(ctx, 1);
  ctx->counters.esdf_blocks = ctx->h_state.esdf_blocks;
  uint32_t sweeps = 0;
  // The wavefronts run to their exact fixed points: min_diff_m only gates the TSDF->ESDF copy
  // of phase 1 (see DESIGN.md §ESDF for why the relaxation itself uses strict improvement).
  EsdfCfgDev cr = c;
  cr.min_diff = 0.0f;
  if (ctx->h_state.esdf_blocks || robot_pending) {
    if (ctx->h_state.esdf_raise_any || robot_pending) {
      rc = esdf_phase<VPS>(ctx, e, cr, 0, used, &sweeps);
      if (rc) return rc;
      hipLaunchKernelGGL(k_esdf_rotate_active, grid_for(used), dim3(256), 0, s, e, used, 1);
    }
    tmark(ctx, 3);
    rc = esdf_phase<VPS>(ctx, e, cr, 1, used, &sweeps);
    if (rc) return rc;
    hipLaunchKernelGGL(k_esdf_rotate_active, grid_for(used), dim3(256), 0, s, e, used, 1);
    rc = esdf_phase<VPS>(ctx, e, cr, 2, used, &sweeps);
    if (rc) return rc;
    tmark(ctx, 6);
    if (ctx->h_state.esdf_raise_any || robot_pending) HIP_TRY(hipMemsetAsync(e.raised, 0, nv, s));
  }
  if (clear_updated_flag && !batch)
    hipLaunchKernelGGL(k_esdf_clear_tsdf_bit, grid_for(used), dim3(256), 0, s, m, e, used);
  tmark(ctx, 7);
  rc = sync_state(ctx);
  if (rc) return rc;
  ctx->counters.esdf_sweeps = sweeps;
  ctx->counters.esdf_relaxations = ctx->h_state.esdf_relax_blocks;
  if (ctx->timing) {
    (void)hipEventSynchronize(ctx->ev[7]);
    vbx_timing& o = ctx->last_timing;
    o = vbx_timing{};
    float t = 0;
    (void)hipEventElapsedTime(&o.total_ms, ctx->ev[0], ctx->ev[7]);
    (void)hipEventElapsedTime(&t, ctx->ev[0], ctx->ev[1]);
    o.prep_ms = t;  // phase 1 (classification)
    if (ctx->ev_hit[3]) { (void)hipEventElapsedTime(&t, ctx->ev[1], ctx->ev[3]); o.solve_ms = t; }  // raise
    if (ctx->ev_hit[6]) { (void)hipEventElapsedTime(&t, ctx->ev[3], ctx->ev[6]); o.fold_ms = t; }   // lower
  }
  return VBX_OK;
}

// EsdfIntegrator::addNewRobotPosition (esdf_integrator.cc:25-92).
int esdf_add_new_robot_position(vbx_ctx* ctx, const vbx_esdf_cfg* cfg, const float position[3]) {
  HIP_TRY(hipSetDevice(ctx->device));
  MapDev& m = ctx->map;
  hipStream_t s = ctx->stream;
  int rc = esdf_ensure(ctx);
  if (rc) return rc;
  EsdfDev e = esdf_dev(ctx);
  HIP_TRY(hipMemsetAsync(&ctx->d_state->new_count, 0, 4, s));
  HIP_TRY(hipMemsetAsync(&ctx->d_state->error, 0, 4, s));
  const float radii[2] = {cfg->clear_sphere_radius, cfg->occupied_sphere_radius};
  for (int pass = 0; pass < 2; ++pass) {
    SphereDev sp;
    // planning_utils_inl.h:18-26: the float loop variable, stepped exactly like the reference's
    sp.r = radii[pass] / m.voxel_size;
    std::vector<float> xs;
    for (float x = -sp.r; x <= sp.r; x++) {
      xs.push_back(x);
      if (xs.size() > 2048) {
        ctx->fail("addNewRobotPosition: sphere radius of more than 1024 voxels");
        return VBX_ERR_INVALID;
      }
    }
    if (xs.empty()) continue;  // negative / NaN radius: empty list
    sp.n = (int)xs.size();
    sp.center = grid_index_from_point(f3{position[0], position[1], position[2]}, m.voxel_size_inv);
    DBuf& bx = pass ? ctx->b_sphere1 : ctx->b_sphere0;
    HIP_TRY(bx.ensure(xs.size() * 4));
    HIP_TRY(hipMemcpyAsync(bx.p, xs.data(), xs.size() * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));  // xs is a stack-lifetime staging buffer
    sp.xs = bx.as<float>();
    const size_t cube = (size_t)sp.n * sp.n * sp.n;
    hipLaunchKernelGGL(k_sphere_mark_blocks, grid_for(cube), dim3(256), 0, s, m, sp, ctx->b_newlist.as<uint32_t>(),
                       ctx->d_state);
    hipLaunchKernelGGL(k_assign_slots, grid_for(m.cap_blocks), dim3(256), 0, s, m, ctx->b_newlist.as<uint32_t>(),
                       ctx->d_state);
    hipLaunchKernelGGL(k_commit_alloc, dim3(1), dim3(1), 0, s, m, ctx->d_state);
    hipLaunchKernelGGL(k_sphere_apply, grid_for(cube), dim3(256), 0, s, m, e, sp, cfg->default_distance_m, pass);
  }
  ctx->esdf_robot_pending = true;
  rc = sync_state(ctx);
  if (rc) return rc;
  return check_state_error(ctx);
}

int esdf_update(vbx_ctx* ctx, const vbx_esdf_cfg* cfg, int batch, int clear_updated_flag,
                const int32_t* list = nullptr, size_t n_list = 0, int list_incremental = 0) {
  HIP_TRY(hipSetDevice(ctx->device));
  if (cfg->full_euclidean_distance) {
    ctx->fail("ESDF: full_euclidean_distance is not supported yet (quasi-Euclidean only)");
    return VBX_ERR_UNSUPPORTED;
  }
  switch (ctx->map.vps) {
    case 8: return esdf_update_t<8>(ctx, cfg, batch, clear_updated_flag, list, n_list, list_incremental);
    case 16: return esdf_update_t<16>(ctx, cfg, batch, clear_updated_flag, list, n_list, list_incremental);
    default:
      ctx->fail("ESDF: voxels_per_side must be 8 or 16 (LDS tile)");
      return VBX_ERR_UNSUPPORTED;
  }
}

}  // namespace

// ---------------------------------------------------------------------------
// C-ABI
// ---------------------------------------------------------------------------
extern "C" {

void vbx_tsdf_cfg_default(vbx_tsdf_cfg* c) {  // tsdf_integrator.h:59-86
  c->default_truncation_distance = 0.1f;
  c->max_weight = 10000.0f;
  c->voxel_carving_enabled = 1;
  c->min_ray_length_m = 0.1f;
  c->max_ray_length_m = 5.0f;
  c->use_const_weight = 0;
  c->allow_clear = 1;
  c->use_weight_dropoff = 1;
  c->use_sparsity_compensation_factor = 0;
  c->sparsity_compensation_factor = 1.0f;
  c->integrator_threads = 1;
  c->integration_order_mode = 0;
  c->enable_anti_grazing = 0;
  c->start_voxel_subsampling_factor = 2.0f;
  c->max_consecutive_ray_collisions = 2;
  c->clear_checks_every_n_frames = 1;
  c->max_integration_time_s = 3.402823466e+38f;
  c->merged_bundle_order = 0;
  c->fast_observed_set = 0;
}

void vbx_esdf_cfg_default(vbx_esdf_cfg* c) {  // esdf_integrator.h:37-77
  c->full_euclidean_distance = 0;
  c->max_distance_m = 2.0f;
  c->min_distance_m = 0.2f;
  c->default_distance_m = 2.0f;
  c->min_diff_m = 0.001f;
  c->min_weight = 1e-6f;
  c->num_buckets = 20;
  c->multi_queue = 0;
  c->add_occupied_crust = 0;
  c->clear_sphere_radius = 1.5f;
  c->occupied_sphere_radius = 5.0f;
}

const char* vbx_last_error(vbx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

vbx_ctx* vbx_create(const vbx_map_cfg* cfg, int device) {
  auto bail = [](vbx_ctx* c, const std::string& msg) -> vbx_ctx* {
    g_create_error = msg;
    if (c) vbx_destroy(c);
    return nullptr;
  };
  if (!cfg || !(cfg->voxel_size > 0.0f)) return bail(nullptr, "vbx_create: voxel_size must be > 0");
  const uint32_t vps = cfg->voxels_per_side;
  if (vps < 4 || vps > 32 || (vps & (vps - 1))) return bail(nullptr, "vbx_create: voxels_per_side must be a power of two in [4,32]");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return bail(nullptr, "vbx_create: no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return bail(nullptr, "vbx_create: bad device ordinal");
  if (hipSetDevice(device) != hipSuccess) return bail(nullptr, "vbx_create: hipSetDevice failed");

  vbx_ctx* ctx = new vbx_ctx;
  ctx->device = device;
  ctx->mcfg = *cfg;
  if (ctx->mcfg.max_blocks == 0) ctx->mcfg.max_blocks = 65536;
  MapDev& m = ctx->map;
  m.cap_blocks = ctx->mcfg.max_blocks;
  m.vps = (int)vps;
  m.vps_log2 = (int)bits_for(vps) - 1;
  m.nvox = vps * vps * vps;
  if ((uint64_t)m.cap_blocks * m.nvox >= (1ull << 32)) {
    return bail(ctx, "vbx_create: max_blocks * voxels_per_side^3 must be < 2^32");
  }
  // Derived constants exactly as Layer's constructor / TsdfIntegratorBase::setLayer compute
  // them (layer.h:34-44, tsdf_integrator.cc:73-79): double division, rounded to float.
  m.voxel_size = cfg->voxel_size;
  m.voxel_size_inv = (float)(1.0 / (double)cfg->voxel_size);
  m.vps_inv = (float)(1.0 / (double)vps);
  uint32_t hcap = 1;
  while (hcap < 4u * m.cap_blocks) hcap <<= 1;
  ctx->hcap = hcap;
  m.hmask = hcap - 1;

  auto ok = [&](hipError_t e) { return e == hipSuccess; };
  const size_t nv = (size_t)m.cap_blocks * m.nvox;
  if (!ok(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)))
    return bail(ctx, "vbx_create: hipStreamCreate failed");
  ctx->stream = ctx->own_stream;
  if (!ok(ctx->b_hkeys.ensure((size_t)hcap * 8)) || !ok(ctx->b_hvals.ensure((size_t)hcap * 4)) ||
      !ok(ctx->b_dist.ensure(nv * 4)) || !ok(ctx->b_weight.ensure(nv * 4)) ||
      !ok(ctx->b_rgba.ensure(nv * 4)) || !ok(ctx->b_blkidx.ensure((size_t)m.cap_blocks * 12)) ||
      !ok(ctx->b_blkflags.ensure((size_t)m.cap_blocks * 4)) ||
      !ok(ctx->b_freelist.ensure((size_t)m.cap_blocks * 4)) ||
      !ok(ctx->b_newlist.ensure((size_t)m.cap_blocks * 4)) ||
      !ok(hipMalloc((void**)&ctx->d_state, sizeof(DevState))))
    return bail(ctx, "vbx_create: out of device memory for the block pool");
  {  // host-visible state mirror (optional: VBX_NO_MIRROR=1 or a failed allocation fall back to copies)
    void* hp = nullptr;
    void* dp = nullptr;
    if (!getenv("VBX_NO_MIRROR") &&
        hipHostMalloc(&hp, sizeof(StateMirror), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
      if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
        std::memset(hp, 0, sizeof(StateMirror));
        ctx->h_mirror = static_cast<StateMirror*>(hp);
        ctx->d_mirror = static_cast<StateMirror*>(dp);
      } else {
        (void)hipHostFree(hp);
      }
    }
    (void)hipGetLastError();
  }
  m.hkeys = ctx->b_hkeys.as<uint64_t>();
  m.hvals = ctx->b_hvals.as<uint32_t>();
  m.dist = ctx->b_dist.as<float>();
  m.weight = ctx->b_weight.as<float>();
  m.rgba = ctx->b_rgba.as<uint32_t>();
  m.blk_idx = ctx->b_blkidx.as<int32_t>();
  m.blk_flags = ctx->b_blkflags.as<uint32_t>();
  m.free_list = ctx->b_freelist.as<uint32_t>();
  hipStream_t s = ctx->stream;
  bool good = ok(hipMemsetAsync(m.hkeys, 0xFF, (size_t)hcap * 8, s)) &&
              ok(hipMemsetAsync(m.hvals, 0xFF, (size_t)hcap * 4, s)) &&
              ok(hipMemsetAsync(m.dist, 0, nv * 4, s)) && ok(hipMemsetAsync(m.weight, 0, nv * 4, s)) &&
              ok(hipMemsetAsync(m.rgba, 0, nv * 4, s)) &&
              ok(hipMemsetAsync(m.blk_flags, 0, (size_t)m.cap_blocks * 4, s)) &&
              ok(hipMemsetAsync(ctx->d_state, 0, sizeof(DevState), s));
  for (int i = 0; i < 9 && good; ++i) good = ok(hipEventCreate(&ctx->ev[i]));
  good = good && ok(hipStreamSynchronize(s));
  if (!good) return bail(ctx, "vbx_create: device initialisation failed");
  return ctx;
}

void vbx_destroy(vbx_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
  DBuf* bufs[] = {&ctx->b_hkeys, &ctx->b_hvals, &ctx->b_dist, &ctx->b_weight, &ctx->b_rgba,
                  &ctx->b_blkidx, &ctx->b_blkflags, &ctx->b_freelist, &ctx->b_newlist, &ctx->b_pts,
                  &ctx->b_cols, &ctx->t_px, &ctx->t_py, &ctx->t_pz, &ctx->t_rgba, &ctx->t_w,
                  &ctx->t_flags, &ctx->t_bkey, &ctx->u_px, &ctx->u_py, &ctx->u_pz, &ctx->u_rgba,
                  &ctx->u_w, &ctx->u_flags, &ctx->u_bkey, &ctx->b_pcx, &ctx->b_pcy, &ctx->b_pcz,
                  &ctx->b_cnt, &ctx->b_off, &ctx->b_keys0, &ctx->b_keys1, &ctx->b_vals0,
                  &ctx->b_vals1, &ctx->b_tmp, &ctx->b_head, &ctx->b_rank, &ctx->b_graze, &ctx->b_T,
                  &ctx->b_U, &ctx->b_vox, &ctx->b_TH, &ctx->b_cl, &ctx->b_act0, &ctx->b_act1, &ctx->b_long, &ctx->b_order, &ctx->b_obs, &ctx->b_sphere0, &ctx->b_sphere1, &ctx->b_redo, &ctx->b_bkeys, &ctx->b_bfirst, &ctx->b_bperm, &ctx->b_obsset, &ctx->b_collided, &ctx->b_startset, &ctx->b_own0, &ctx->b_own1, &ctx->b_edist, &ctx->b_estate,
                  &ctx->b_eraised, &ctx->b_eactive};
  for (DBuf* b : bufs) b->release();
  if (ctx->d_state) (void)hipFree(ctx->d_state);
  if (ctx->h_mirror) (void)hipHostFree(ctx->h_mirror);
  for (int i = 0; i < 9; ++i)
    if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

int vbx_set_stream(vbx_ctx* ctx, void* hip_stream) {
  if (!ctx) return VBX_ERR_INVALID;
  ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
  return VBX_OK;
}

int vbx_tsdf_integrate_device(vbx_ctx* ctx, int kind, const vbx_tsdf_cfg* cfg, const float pos[3],
                              const float quat[4], const float* d_points_C, const uint8_t* d_rgba,
                              size_t n, int freespace_points) {
  if (!ctx) return VBX_ERR_INVALID;
  return integrate_device(ctx, kind, cfg, pos, quat, d_points_C, d_rgba, n, freespace_points);
}

int vbx_tsdf_integrate(vbx_ctx* ctx, int kind, const vbx_tsdf_cfg* cfg, const float pos[3],
                       const float quat[4], const float* points_C, const uint8_t* rgba, size_t n,
                       int freespace_points) {
  if (!ctx) return VBX_ERR_INVALID;
  if (n && (!points_C || !rgba)) {
    ctx->fail("vbx_tsdf_integrate: null input");
    return VBX_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  if (n) {
    HIP_TRY(ctx->b_pts.ensure(n * 12));
    HIP_TRY(ctx->b_cols.ensure(n * 4));
    HIP_TRY(hipMemcpyAsync(ctx->b_pts.p, points_C, n * 12, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->b_cols.p, rgba, n * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  return integrate_device(ctx, kind, cfg, pos, quat, ctx->b_pts.as<float>(),
                          ctx->b_cols.as<uint8_t>(), n, freespace_points);
}

int vbx_esdf_update(vbx_ctx* ctx, const vbx_esdf_cfg* cfg, int batch, int clear_updated_flag) {
  if (!ctx || !cfg) return VBX_ERR_INVALID;
  return esdf_update(ctx, cfg, batch, clear_updated_flag);
}

int vbx_esdf_update_blocks(vbx_ctx* ctx, const vbx_esdf_cfg* cfg, const int32_t* idx_xyz, size_t n, int incremental) {
  if (!ctx || !cfg || (n && !idx_xyz)) return VBX_ERR_INVALID;
  static const int32_t kNone[3] = {0, 0, 0};
  return esdf_update(ctx, cfg, 0, 0, n ? idx_xyz : kNone, n, incremental);
}

int vbx_esdf_integrator_clear(vbx_ctx* ctx) {
  if (!ctx) return VBX_ERR_INVALID;
  HIP_TRY(hipSetDevice(ctx->device));
  ctx->esdf_robot_pending = false;
  if (!ctx->esdf_init) return VBX_OK;
  int rc = sync_state(ctx);
  if (rc) return rc;
  const uint32_t used = ctx->h_state.pool_used;
  if (used == 0) return VBX_OK;
  HIP_TRY(hipMemsetAsync(ctx->b_eraised.p, 0, (size_t)used * ctx->map.nvox, ctx->stream));
  hipLaunchKernelGGL(k_clear_update_bits, grid_for(used), dim3(256), 0, ctx->stream, ctx->map, used, ~0u,
                     kFlagEsdfPendClassify | kFlagEsdfPendOpen);
  return VBX_OK;
}

int vbx_esdf_add_new_robot_position(vbx_ctx* ctx, const vbx_esdf_cfg* cfg, const float position[3]) {
  if (!ctx || !cfg || !position) return VBX_ERR_INVALID;
  return esdf_add_new_robot_position(ctx, cfg, position);
}

// ---- block listing / transfer --------------------------------------------------------
static int list_blocks(vbx_ctx* ctx, int layer, uint32_t need_mask, std::vector<std::pair<uint64_t, uint32_t>>* out) {
  if (layer != VBX_LAYER_TSDF && layer != VBX_LAYER_ESDF) {
    ctx->fail("unknown layer %d", layer);
    return VBX_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  int rc = sync_state(ctx);
  if (rc) return rc;
  const uint32_t used = ctx->h_state.pool_used;
  std::vector<uint32_t> flags(used);
  std::vector<int32_t> idx((size_t)used * 3);
  if (used) {
    HIP_TRY(hipMemcpy(flags.data(), ctx->map.blk_flags, (size_t)used * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(idx.data(), ctx->map.blk_idx, (size_t)used * 12, hipMemcpyDeviceToHost));
  }
  out->clear();
  for (uint32_t sl = 0; sl < used; ++sl) {
    if (layer == VBX_LAYER_ESDF) {
      if (!(flags[sl] & kFlagEsdfAlloc)) continue;
      if (need_mask && !((flags[sl] >> kFlagEsdfUpdShift) & need_mask)) continue;
    } else {
      if (!(flags[sl] & kFlagPublished)) continue;
      if (need_mask && !(flags[sl] & need_mask)) continue;
    }
    out->emplace_back(pack_block_key(idx[3 * sl], idx[3 * sl + 1], idx[3 * sl + 2]), sl);
  }
  std::sort(out->begin(), out->end());
  return VBX_OK;
}

int vbx_num_blocks(vbx_ctx* ctx, int layer, size_t* n) {
  if (!ctx || !n) return VBX_ERR_INVALID;
  std::vector<std::pair<uint64_t, uint32_t>> v;
  int rc = list_blocks(ctx, layer, 0, &v);
  if (rc) return rc;
  *n = v.size();
  return VBX_OK;
}

static int emit_list(const std::vector<std::pair<uint64_t, uint32_t>>& v, int32_t* idx, size_t cap, size_t* n) {
  *n = v.size();
  for (size_t i = 0; i < v.size() && i < cap; ++i) {
    int x, y, z;
    unpack_block_key(v[i].first, &x, &y, &z);
    idx[3 * i] = x; idx[3 * i + 1] = y; idx[3 * i + 2] = z;
  }
  return VBX_OK;
}

int vbx_block_indices(vbx_ctx* ctx, int layer, int32_t* idx, size_t cap, size_t* n) {
  if (!ctx || !n || (cap && !idx)) return VBX_ERR_INVALID;
  std::vector<std::pair<uint64_t, uint32_t>> v;
  int rc = list_blocks(ctx, layer, 0, &v);
  if (rc) return rc;
  return emit_list(v, idx, cap, n);
}

int vbx_blocks_updated(vbx_ctx* ctx, int layer, int update_mask, int32_t* idx, size_t cap, size_t* n) {
  if (!ctx || !n || (cap && !idx)) return VBX_ERR_INVALID;
  std::vector<std::pair<uint64_t, uint32_t>> v;
  int rc = list_blocks(ctx, layer, (uint32_t)update_mask & kFlagUpdMask, &v);
  if (rc) return rc;
  return emit_list(v, idx, cap, n);
}

static int find_slot_host(vbx_ctx* ctx, const int32_t idx[3], uint32_t* slot, uint32_t* hpos) {
  // host-side probe of the device hash map (small D2H reads; not on the hot path)
  const uint64_t key = pack_block_key(idx[0], idx[1], idx[2]);
  uint32_t h = mix_key(key) & ctx->map.hmask;
  for (uint32_t probes = 0; probes <= ctx->map.hmask; ++probes) {
    uint64_t k;
    HIP_TRY(hipMemcpy(&k, ctx->map.hkeys + h, 8, hipMemcpyDeviceToHost));
    if (k == key) {
      HIP_TRY(hipMemcpy(slot, ctx->map.hvals + h, 4, hipMemcpyDeviceToHost));
      if (hpos) *hpos = h;
      return VBX_OK;
    }
    if (k == kEmptyKey) break;
    h = (h + 1) & ctx->map.hmask;
  }
  *slot = kInvalidSlot;
  return VBX_OK;
}

int vbx_block_download(vbx_ctx* ctx, int layer, const int32_t idx[3], void* aos, uint8_t* updated_bits,
                       uint8_t* has_data) {
  if (!ctx || !idx || !aos) return VBX_ERR_INVALID;
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  uint32_t slot;
  int rc = find_slot_host(ctx, idx, &slot, nullptr);
  if (rc) return rc;
  uint32_t flags = 0;
  if (slot != kInvalidSlot) HIP_TRY(hipMemcpy(&flags, ctx->map.blk_flags + slot, 4, hipMemcpyDeviceToHost));
  const uint32_t need = (layer == VBX_LAYER_ESDF) ? kFlagEsdfAlloc : kFlagPublished;
  if (slot == kInvalidSlot || !(flags & need) || (layer == VBX_LAYER_ESDF && !ctx->esdf_init)) {
    ctx->fail("block (%d,%d,%d) is not allocated", idx[0], idx[1], idx[2]);
    return VBX_ERR_INVALID;
  }
  const uint32_t nv = ctx->map.nvox;
  if (layer == VBX_LAYER_ESDF) {
    std::vector<float> d(nv);
    std::vector<uint32_t> st(nv);
    HIP_TRY(hipMemcpy(d.data(), ctx->b_edist.as<float>() + (size_t)slot * nv, nv * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(st.data(), ctx->b_estate.as<uint32_t>() + (size_t)slot * nv, nv * 4, hipMemcpyDeviceToHost));
    uint8_t* o = static_cast<uint8_t*>(aos);
    for (uint32_t i = 0; i < nv; ++i) {  // EsdfVoxel AoS, voxel.h:18-37 (20 bytes)
      std::memcpy(o + 20 * i, &d[i], 4);
      o[20 * i + 4] = (st[i] & 1) ? 1 : 0;
      o[20 * i + 5] = (st[i] & 2) ? 1 : 0;
      o[20 * i + 6] = (st[i] & 4) ? 1 : 0;
      o[20 * i + 7] = (st[i] & 8) ? 1 : 0;
      const int32_t p[3] = {(int32_t)(int8_t)((st[i] >> 8) & 0xFF), (int32_t)(int8_t)((st[i] >> 16) & 0xFF),
                            (int32_t)(int8_t)((st[i] >> 24) & 0xFF)};
      std::memcpy(o + 20 * i + 8, p, 12);
    }
    if (updated_bits) *updated_bits = (uint8_t)((flags >> kFlagEsdfUpdShift) & kFlagUpdMask);
    if (has_data) *has_data = 0;
    return VBX_OK;
  }
  std::vector<float> d(nv), w(nv);
  std::vector<uint32_t> c(nv);
  HIP_TRY(hipMemcpy(d.data(), ctx->map.dist + (size_t)slot * nv, nv * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(w.data(), ctx->map.weight + (size_t)slot * nv, nv * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(c.data(), ctx->map.rgba + (size_t)slot * nv, nv * 4, hipMemcpyDeviceToHost));
  uint8_t* o = static_cast<uint8_t*>(aos);
  for (uint32_t i = 0; i < nv; ++i) {  // TsdfVoxel AoS, voxel.h:12-16
    std::memcpy(o + 12 * i, &d[i], 4);
    std::memcpy(o + 12 * i + 4, &w[i], 4);
    std::memcpy(o + 12 * i + 8, &c[i], 4);
  }
  if (updated_bits) *updated_bits = (uint8_t)(flags & kFlagUpdMask);
  if (has_data) *has_data = (flags & kFlagHasData) ? 1 : 0;
  return VBX_OK;
}

void* vbx_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}
void vbx_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

static int upload_idx(vbx_ctx* ctx, const int32_t* idx, size_t n);

int vbx_blocks_download(vbx_ctx* ctx, int layer, const int32_t* idx, size_t n, void* aos, uint8_t* updated_bits,
                        uint8_t* has_data) {
  if (!ctx || (n && (!idx || !aos))) return VBX_ERR_INVALID;
  if (layer != VBX_LAYER_TSDF && layer != VBX_LAYER_ESDF) {
    ctx->fail("unknown layer %d", layer);
    return VBX_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  if (n == 0) return VBX_OK;
  if (layer == VBX_LAYER_ESDF && !ctx->esdf_init) {
    ctx->fail("ESDF layer is empty");
    return VBX_ERR_INVALID;
  }
  int rc = upload_idx(ctx, idx, n);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  MapDev& m = ctx->map;
  const size_t wpb = (size_t)m.nvox * (layer == VBX_LAYER_TSDF ? 3 : 5);
  HIP_TRY(ctx->b_keys0.ensure(n * wpb * 4));
  HIP_TRY(ctx->b_vals0.ensure(n * 4));
  hipLaunchKernelGGL(k_lookup_slots_flags, grid_for(n), dim3(256), 0, s, m, ctx->b_head.as<int32_t>(), (uint32_t)n,
                     layer == VBX_LAYER_TSDF ? kFlagPublished : kFlagEsdfAlloc, ctx->b_rank.as<uint32_t>(),
                     ctx->b_vals0.as<uint32_t>());
  if (layer == VBX_LAYER_TSDF)
    hipLaunchKernelGGL(k_pack_tsdf_aos, dim3((unsigned)n), dim3(256), 0, s, m, ctx->b_rank.as<uint32_t>(),
                       ctx->b_keys0.as<uint32_t>());
  else
    hipLaunchKernelGGL(k_pack_esdf_aos, dim3((unsigned)n), dim3(256), 0, s, m.nvox, ctx->b_edist.as<float>(),
                       ctx->b_estate.as<uint32_t>(), ctx->b_rank.as<uint32_t>(), ctx->b_keys0.as<uint32_t>());
  std::vector<uint32_t> flags(n);
  HIP_TRY(hipMemcpyAsync(flags.data(), ctx->b_vals0.p, n * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(aos, ctx->b_keys0.p, n * wpb * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (size_t i = 0; i < n; ++i) {
    if (flags[i] == ~0u) {
      ctx->fail("block (%d,%d,%d) is not allocated", idx[3 * i], idx[3 * i + 1], idx[3 * i + 2]);
      return VBX_ERR_INVALID;
    }
    if (layer == VBX_LAYER_TSDF) {
      if (updated_bits) updated_bits[i] = (uint8_t)(flags[i] & kFlagUpdMask);
      if (has_data) has_data[i] = (flags[i] & kFlagHasData) ? 1 : 0;
    } else {
      if (updated_bits) updated_bits[i] = (uint8_t)((flags[i] >> kFlagEsdfUpdShift) & kFlagUpdMask);
      if (has_data) has_data[i] = 0;
    }
  }
  return VBX_OK;
}

int vbx_block_upload(vbx_ctx* ctx, int layer, const int32_t idx[3], const void* aos, uint8_t updated_bits,
                     uint8_t has_data) {
  if (!ctx || !idx || !aos) return VBX_ERR_INVALID;
  if (layer != VBX_LAYER_TSDF && layer != VBX_LAYER_ESDF) {
    ctx->fail("unknown layer %d", layer);
    return VBX_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  int rc = VBX_OK;
  if (layer == VBX_LAYER_ESDF) {
    rc = esdf_ensure(ctx);
    if (rc) return rc;
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  uint32_t slot;
  rc = find_slot_host(ctx, idx, &slot, nullptr);
  if (rc) return rc;
  MapDev& m = ctx->map;
  uint32_t old_flags = 0;
  if (slot != kInvalidSlot) HIP_TRY(hipMemcpy(&old_flags, m.blk_flags + slot, 4, hipMemcpyDeviceToHost));
  if (slot == kInvalidSlot) {
    // host-side insert: take a slot from the free list or the bump pointer
    rc = sync_state(ctx);
    if (rc) return rc;
    if (ctx->h_state.free_count > 0) {
      HIP_TRY(hipMemcpy(&slot, m.free_list + (ctx->h_state.free_count - 1), 4, hipMemcpyDeviceToHost));
      ctx->h_state.free_count--;
    } else {
      if (ctx->h_state.pool_used >= m.cap_blocks) {
        ctx->fail("block pool full");
        return VBX_ERR_CAPACITY;
      }
      slot = ctx->h_state.pool_used++;
    }
    HIP_TRY(hipMemcpy(&ctx->d_state->pool_used, &ctx->h_state.pool_used, 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(&ctx->d_state->free_count, &ctx->h_state.free_count, 4, hipMemcpyHostToDevice));
    const uint64_t key = pack_block_key(idx[0], idx[1], idx[2]);
    uint32_t h = mix_key(key) & m.hmask;
    for (;;) {
      uint64_t k;
      HIP_TRY(hipMemcpy(&k, m.hkeys + h, 8, hipMemcpyDeviceToHost));
      if (k == kEmptyKey) break;
      h = (h + 1) & m.hmask;
    }
    HIP_TRY(hipMemcpy(m.hkeys + h, &key, 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(m.hvals + h, &slot, 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(m.blk_idx + 3 * slot, idx, 12, hipMemcpyHostToDevice));
  }
  const uint32_t nv = m.nvox;
  const uint32_t esdf_bits = kFlagEsdfAlloc | (kFlagUpdMask << kFlagEsdfUpdShift) | kFlagEsdfPendClassify | kFlagEsdfPendOpen;
  if (layer == VBX_LAYER_ESDF) {  // EsdfVoxel AoS (voxel.h:18-37) -> distance + packed state
    std::vector<float> d(nv);
    std::vector<uint32_t> st(nv);
    const uint8_t* in = static_cast<const uint8_t*>(aos);
    for (uint32_t i = 0; i < nv; ++i) {
      std::memcpy(&d[i], in + 20 * i, 4);
      int32_t par[3];
      std::memcpy(par, in + 20 * i + 8, 12);
      st[i] = (in[20 * i + 4] ? 1u : 0u) | (in[20 * i + 5] ? 2u : 0u) | (in[20 * i + 6] ? 4u : 0u) |
              (in[20 * i + 7] ? 8u : 0u) | ((uint32_t)(uint8_t)(int8_t)par[0] << 8) |
              ((uint32_t)(uint8_t)(int8_t)par[1] << 16) | ((uint32_t)(uint8_t)(int8_t)par[2] << 24);
    }
    HIP_TRY(hipMemcpy(ctx->b_edist.as<float>() + (size_t)slot * nv, d.data(), nv * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->b_estate.as<uint32_t>() + (size_t)slot * nv, st.data(), nv * 4, hipMemcpyHostToDevice));
    const uint32_t flags = (old_flags & ~esdf_bits) | kFlagEsdfAlloc |
                           (((uint32_t)updated_bits & kFlagUpdMask) << kFlagEsdfUpdShift);
    HIP_TRY(hipMemcpy(m.blk_flags + slot, &flags, 4, hipMemcpyHostToDevice));
    return VBX_OK;
  }
  std::vector<float> d(nv), w(nv);
  std::vector<uint32_t> c(nv);
  const uint8_t* in = static_cast<const uint8_t*>(aos);
  for (uint32_t i = 0; i < nv; ++i) {
    std::memcpy(&d[i], in + 12 * i, 4);
    std::memcpy(&w[i], in + 12 * i + 4, 4);
    std::memcpy(&c[i], in + 12 * i + 8, 4);
  }
  HIP_TRY(hipMemcpy(m.dist + (size_t)slot * nv, d.data(), nv * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(m.weight + (size_t)slot * nv, w.data(), nv * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(m.rgba + (size_t)slot * nv, c.data(), nv * 4, hipMemcpyHostToDevice));
  // the ESDF layer's membership of the same slot is untouched (the layers are independent)
  const uint32_t flags = (old_flags & esdf_bits) | kFlagPublished | (updated_bits & kFlagUpdMask) | (has_data ? kFlagHasData : 0);
  HIP_TRY(hipMemcpy(m.blk_flags + slot, &flags, 4, hipMemcpyHostToDevice));
  return VBX_OK;
}

static int remove_slot(vbx_ctx* ctx, int layer, uint32_t slot, uint32_t hpos) {
  // Unpublish and zero the block; the hash entry stays and keeps its slot, so the block is
  // simply a zeroed "candidate" again (no tombstones, no free-list churn).
  MapDev& m = ctx->map;
  (void)hpos;
  const uint32_t nv = m.nvox;
  const uint32_t zero = 0;
  if (layer == VBX_LAYER_ESDF) {
    if (!ctx->esdf_init) return VBX_OK;
    uint32_t f;
    HIP_TRY(hipMemset(ctx->b_edist.as<float>() + (size_t)slot * nv, 0, nv * 4));
    HIP_TRY(hipMemset(ctx->b_estate.as<uint32_t>() + (size_t)slot * nv, 0, nv * 4));
    HIP_TRY(hipMemcpy(&f, m.blk_flags + slot, 4, hipMemcpyDeviceToHost));
    f &= ~(kFlagEsdfAlloc | (kFlagUpdMask << kFlagEsdfUpdShift));
    HIP_TRY(hipMemcpy(m.blk_flags + slot, &f, 4, hipMemcpyHostToDevice));
    return VBX_OK;
  }
  {  // a removed TSDF block keeps its ESDF flags (the layers are independent, layer.h:167)
    uint32_t f;
    HIP_TRY(hipMemcpy(&f, m.blk_flags + slot, 4, hipMemcpyDeviceToHost));
    f &= (kFlagEsdfAlloc | (kFlagUpdMask << kFlagEsdfUpdShift));
    HIP_TRY(hipMemset(m.dist + (size_t)slot * nv, 0, nv * 4));
    HIP_TRY(hipMemset(m.weight + (size_t)slot * nv, 0, nv * 4));
    HIP_TRY(hipMemset(m.rgba + (size_t)slot * nv, 0, nv * 4));
    HIP_TRY(hipMemcpy(m.blk_flags + slot, &f, 4, hipMemcpyHostToDevice));
    return VBX_OK;
  }
  HIP_TRY(hipMemset(m.dist + (size_t)slot * nv, 0, nv * 4));
  HIP_TRY(hipMemset(m.weight + (size_t)slot * nv, 0, nv * 4));
  HIP_TRY(hipMemset(m.rgba + (size_t)slot * nv, 0, nv * 4));
  HIP_TRY(hipMemcpy(m.blk_flags + slot, &zero, 4, hipMemcpyHostToDevice));
  return VBX_OK;
}

int vbx_block_remove(vbx_ctx* ctx, int layer, const int32_t idx[3]) {
  if (!ctx || !idx) return VBX_ERR_INVALID;
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  uint32_t slot, hpos = 0;
  int rc = find_slot_host(ctx, idx, &slot, &hpos);
  if (rc) return rc;
  if (slot == kInvalidSlot) return VBX_OK;  // unordered_map::erase of a missing key is a no-op
  return remove_slot(ctx, layer, slot, hpos);
}

int vbx_remove_distant_blocks(vbx_ctx* ctx, int layer, const float center[3], double max_distance) {
  if (!ctx || !center) return VBX_ERR_INVALID;
  if (layer != VBX_LAYER_TSDF && layer != VBX_LAYER_ESDF) {
    ctx->fail("unknown layer %d", layer);
    return VBX_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  int rc = sync_state(ctx);
  if (rc) return rc;
  const uint32_t used = ctx->h_state.pool_used;
  if (used == 0) return VBX_OK;
  const float block_size = ctx->map.voxel_size * (float)ctx->map.vps;
  hipLaunchKernelGGL(k_remove_distant, dim3(used), dim3(256), 0, ctx->stream, ctx->map,
                     ctx->esdf_init ? ctx->b_edist.as<float>() : (float*)nullptr,
                     ctx->esdf_init ? ctx->b_estate.as<uint32_t>() : (uint32_t*)nullptr, layer,
                     f3{center[0], center[1], center[2]}, max_distance * max_distance, block_size);
  return VBX_OK;
}

int vbx_clear(vbx_ctx* ctx, int layer) {
  if (!ctx) return VBX_ERR_INVALID;
  HIP_TRY(hipSetDevice(ctx->device));
  {
    // removeAllBlocks (layer.h:168): every block of the layer is zeroed and leaves it in one
    // launch (a workgroup per pool slot, slots outside the layer exit at once); the hash entries
    // stay and turn back into invisible candidates.
    int rc = sync_state(ctx);
    if (rc) return rc;
    const uint32_t used = ctx->h_state.pool_used;
    if (used == 0) return VBX_OK;
    if (layer != VBX_LAYER_TSDF && layer != VBX_LAYER_ESDF) {
      ctx->fail("unknown layer %d", layer);
      return VBX_ERR_INVALID;
    }
    hipLaunchKernelGGL(k_remove_distant, dim3(used), dim3(256), 0, ctx->stream, ctx->map,
                       ctx->esdf_init ? ctx->b_edist.as<float>() : (float*)nullptr,
                       ctx->esdf_init ? ctx->b_estate.as<uint32_t>() : (uint32_t*)nullptr, layer, f3{0.f, 0.f, 0.f},
                       -1.0, 0.0f);  // squared distance > -1: every block
    return VBX_OK;
  }
}

int vbx_clear_updated(vbx_ctx* ctx, int layer, int update_mask) {
  if (!ctx) return VBX_ERR_INVALID;
  if (layer != VBX_LAYER_TSDF && layer != VBX_LAYER_ESDF) {
    ctx->fail("unknown layer %d", layer);
    return VBX_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  int rc = sync_state(ctx);
  if (rc) return rc;
  const uint32_t used = ctx->h_state.pool_used;
  if (used == 0) return VBX_OK;
  const uint32_t bits = (layer == VBX_LAYER_ESDF) ? (((uint32_t)update_mask & kFlagUpdMask) << kFlagEsdfUpdShift)
                                                  : ((uint32_t)update_mask & kFlagUpdMask);
  hipLaunchKernelGGL(k_clear_update_bits, grid_for(used), dim3(256), 0, ctx->stream, ctx->map, used,
                     layer == VBX_LAYER_ESDF ? kFlagEsdfAlloc : kFlagPublished, bits);
  return VBX_OK;
}


// ---- multi-GPU block merge ------------------------------------------------------------
static int upload_idx(vbx_ctx* ctx, const int32_t* idx, size_t n) {
  HIP_TRY(ctx->b_head.ensure(std::max<size_t>(n, 1) * 12));
  HIP_TRY(ctx->b_rank.ensure(std::max<size_t>(n, 1) * 4));
  HIP_TRY(hipMemcpyAsync(ctx->b_head.p, idx, n * 12, hipMemcpyHostToDevice, ctx->stream));
  return VBX_OK;
}

int vbx_blocks_export_sums(vbx_ctx* ctx, const int32_t* idx, size_t n, float* d_out) {
  if (!ctx || (n && (!idx || !d_out))) return VBX_ERR_INVALID;
  HIP_TRY(hipSetDevice(ctx->device));
  if (n == 0) return VBX_OK;
  int rc = upload_idx(ctx, idx, n);
  if (rc) return rc;
  hipLaunchKernelGGL(k_lookup_slots, grid_for(n), dim3(256), 0, ctx->stream, ctx->map,
                     ctx->b_head.as<int32_t>(), (uint32_t)n, 1, ctx->b_rank.as<uint32_t>());
  hipLaunchKernelGGL(k_export_sums, dim3((unsigned)n), dim3(256), 0, ctx->stream, ctx->map,
                     ctx->b_rank.as<uint32_t>(), d_out);
  HIP_TRY(hipStreamSynchronize(ctx->stream));  // idx is a caller-owned host buffer
  return VBX_OK;
}

int vbx_blocks_merge_sums(vbx_ctx* ctx, const int32_t* idx, size_t n, const float* d_sums, int apply_caps,
                          float truncation_distance, float max_weight) {
  if (!ctx || (n && (!idx || !d_sums))) return VBX_ERR_INVALID;
  HIP_TRY(hipSetDevice(ctx->device));
  if (n == 0) return VBX_OK;
  int rc = upload_idx(ctx, idx, n);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  MapDev& m = ctx->map;
  HIP_TRY(hipMemsetAsync(&ctx->d_state->new_count, 0, 4, s));
  HIP_TRY(hipMemsetAsync(&ctx->d_state->error, 0, 4, s));
  hipLaunchKernelGGL(k_insert_blocks, grid_for(n), dim3(256), 0, s, m, ctx->b_head.as<int32_t>(), (uint32_t)n,
                     ctx->b_newlist.as<uint32_t>(), ctx->d_state);
  hipLaunchKernelGGL(k_assign_slots, grid_for(m.cap_blocks), dim3(256), 0, s, m, ctx->b_newlist.as<uint32_t>(),
                     ctx->d_state);
  hipLaunchKernelGGL(k_commit_alloc, dim3(1), dim3(1), 0, s, m, ctx->d_state);
  hipLaunchKernelGGL(k_lookup_slots, grid_for(n), dim3(256), 0, s, m, ctx->b_head.as<int32_t>(), (uint32_t)n, 0,
                     ctx->b_rank.as<uint32_t>());
  hipLaunchKernelGGL(k_merge_sums, dim3((unsigned)n), dim3(256), 0, s, m, ctx->b_rank.as<uint32_t>(), d_sums,
                     apply_caps, truncation_distance, max_weight, ctx->d_state);
  rc = sync_state(ctx);
  if (rc) return rc;
  return check_state_error(ctx);
}


// ---- Layer serialization -----------------------------------------------------------------
int vbx_blocks_serialize(vbx_ctx* ctx, int layer, const int32_t* idx, size_t n, uint32_t* words, uint8_t* has_data) {
  if (!ctx || (n && (!idx || !words))) return VBX_ERR_INVALID;
  if (layer != VBX_LAYER_TSDF && layer != VBX_LAYER_ESDF) return VBX_ERR_INVALID;
  HIP_TRY(hipSetDevice(ctx->device));
  if (n == 0) return VBX_OK;
  if (layer == VBX_LAYER_ESDF && !ctx->esdf_init) {
    ctx->fail("ESDF layer is empty");
    return VBX_ERR_INVALID;
  }
  int rc = upload_idx(ctx, idx, n);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  MapDev& m = ctx->map;
  const size_t wpb = (size_t)m.nvox * (layer == VBX_LAYER_TSDF ? 3 : 2);
  HIP_TRY(ctx->b_keys0.ensure(n * wpb * 4));
  HIP_TRY(ctx->b_vals0.ensure(n * 4));
  hipLaunchKernelGGL(k_lookup_slots_flags, grid_for(n), dim3(256), 0, s, m, ctx->b_head.as<int32_t>(), (uint32_t)n,
                     layer == VBX_LAYER_TSDF ? kFlagPublished : kFlagEsdfAlloc, ctx->b_rank.as<uint32_t>(),
                     ctx->b_vals0.as<uint32_t>());
  if (layer == VBX_LAYER_TSDF)
    hipLaunchKernelGGL(k_serialize_tsdf, dim3((unsigned)n), dim3(256), 0, s, m, ctx->b_rank.as<uint32_t>(),
                       ctx->b_keys0.as<uint32_t>());
  else
    hipLaunchKernelGGL(k_serialize_esdf, dim3((unsigned)n), dim3(256), 0, s, m.nvox, ctx->b_edist.as<float>(),
                       ctx->b_estate.as<uint32_t>(), ctx->b_rank.as<uint32_t>(), ctx->b_keys0.as<uint32_t>());
  std::vector<uint32_t> flags(n);
  HIP_TRY(hipMemcpyAsync(flags.data(), ctx->b_vals0.p, n * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(words, ctx->b_keys0.p, n * wpb * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (size_t i = 0; i < n; ++i) {
    if (flags[i] == ~0u) {
      ctx->fail("block (%d,%d,%d) is not allocated", idx[3 * i], idx[3 * i + 1], idx[3 * i + 2]);
      return VBX_ERR_INVALID;
    }
    if (has_data) has_data[i] = (layer == VBX_LAYER_TSDF && (flags[i] & kFlagHasData)) ? 1 : 0;
  }
  return VBX_OK;
}

int vbx_blocks_deserialize(vbx_ctx* ctx, int layer, const int32_t* idx, size_t n, const uint32_t* words,
                           const uint8_t* has_data) {
  if (!ctx || (n && (!idx || !words))) return VBX_ERR_INVALID;
  if (layer != VBX_LAYER_TSDF && layer != VBX_LAYER_ESDF) return VBX_ERR_INVALID;
  HIP_TRY(hipSetDevice(ctx->device));
  if (n == 0) return VBX_OK;
  int rc = upload_idx(ctx, idx, n);
  if (rc) return rc;
  if (layer == VBX_LAYER_ESDF) {
    rc = esdf_ensure(ctx);
    if (rc) return rc;
  }
  hipStream_t s = ctx->stream;
  MapDev& m = ctx->map;
  const size_t wpb = (size_t)m.nvox * (layer == VBX_LAYER_TSDF ? 3 : 2);
  HIP_TRY(ctx->b_keys0.ensure(n * wpb * 4));
  HIP_TRY(hipMemcpyAsync(ctx->b_keys0.p, words, n * wpb * 4, hipMemcpyHostToDevice, s));
  uint8_t* d_hd = nullptr;
  if (has_data) {
    HIP_TRY(ctx->b_graze.ensure(n));
    HIP_TRY(hipMemcpyAsync(ctx->b_graze.p, has_data, n, hipMemcpyHostToDevice, s));
    d_hd = ctx->b_graze.as<uint8_t>();
  }
  HIP_TRY(hipMemsetAsync(&ctx->d_state->new_count, 0, 4, s));
  HIP_TRY(hipMemsetAsync(&ctx->d_state->error, 0, 4, s));
  hipLaunchKernelGGL(k_insert_blocks, grid_for(n), dim3(256), 0, s, m, ctx->b_head.as<int32_t>(), (uint32_t)n,
                     ctx->b_newlist.as<uint32_t>(), ctx->d_state);
  hipLaunchKernelGGL(k_assign_slots, grid_for(m.cap_blocks), dim3(256), 0, s, m, ctx->b_newlist.as<uint32_t>(),
                     ctx->d_state);
  hipLaunchKernelGGL(k_commit_alloc, dim3(1), dim3(1), 0, s, m, ctx->d_state);
  hipLaunchKernelGGL(k_lookup_slots, grid_for(n), dim3(256), 0, s, m, ctx->b_head.as<int32_t>(), (uint32_t)n, 0,
                     ctx->b_rank.as<uint32_t>());
  if (layer == VBX_LAYER_TSDF) {
    hipLaunchKernelGGL(k_deserialize_tsdf, dim3((unsigned)n), dim3(256), 0, s, m, ctx->b_rank.as<uint32_t>(),
                       ctx->b_keys0.as<uint32_t>());
    hipLaunchKernelGGL(k_set_block_flags, grid_for(n), dim3(256), 0, s, m, ctx->b_rank.as<uint32_t>(), (uint32_t)n,
                       kFlagPublished | kFlagUpdMask, d_hd, kFlagHasData);
  } else {
    hipLaunchKernelGGL(k_deserialize_esdf, dim3((unsigned)n), dim3(256), 0, s, m.nvox, ctx->b_edist.as<float>(),
                       ctx->b_estate.as<uint32_t>(), ctx->b_rank.as<uint32_t>(), ctx->b_keys0.as<uint32_t>());
    hipLaunchKernelGGL(k_set_block_flags, grid_for(n), dim3(256), 0, s, m, ctx->b_rank.as<uint32_t>(), (uint32_t)n,
                       kFlagEsdfAlloc | (kFlagUpdMask << kFlagEsdfUpdShift), (const uint8_t*)nullptr, 0u);
  }
  rc = sync_state(ctx);
  if (rc) return rc;
  return check_state_error(ctx);
}

int vbx_get_counters(vbx_ctx* ctx, vbx_counters* out) {
  if (!ctx || !out) return VBX_ERR_INVALID;
  *out = ctx->counters;
  return VBX_OK;
}
int vbx_enable_timing(vbx_ctx* ctx, int enable) {
  if (!ctx) return VBX_ERR_INVALID;
  ctx->timing = enable != 0;
  return VBX_OK;
}
int vbx_get_timing(vbx_ctx* ctx, vbx_timing* out) {
  if (!ctx || !out) return VBX_ERR_INVALID;
  *out = ctx->last_timing;
  return VBX_OK;
}

}  // extern "C"
