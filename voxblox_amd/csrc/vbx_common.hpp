// vbx_common.hpp — device-visible structs (block pool, hash map, ray tables, state) and the hash map's device functions
// Part of libvbx_hip.so's single translation unit (included by vbx_hip.hip, in order).

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

struct PBuf {  // growable page-locked host buffer (small per-frame read-backs / uploads)
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = (std::max(bytes, cap + cap / 2) + 4095) & ~size_t(4095);
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

constexpr uint64_t kEmptyKey = ~0ull;
constexpr uint32_t kInvalidSlot = 0xFFFFFFFFu;
constexpr uint64_t kTombKey = ~0ull - 1;  // hash entry of a recycled block: lookups walk past it, inserts never claim it

// block flag bits (blk_flags[slot])
constexpr uint32_t kFlagUpdMask = 0x7;      // Update::kMap|kMesh|kEsdf, core/block.h:15-18
constexpr uint32_t kFlagPublished = 0x100;  // block is part of the API-visible Layer
constexpr uint32_t kFlagHasData = 0x200;    // Block::has_data_
constexpr uint32_t kFlagNewThisCall = 0x400;
constexpr uint32_t kFlagEsdfAlloc = 0x1000;   // block exists in Layer<EsdfVoxel>
constexpr uint32_t kFlagEsdfUpdShift = 4;     // ESDF block's Update bits live in bits 4..6
constexpr uint32_t kFlagEsdfPendClassify = 0x2000;  // EsdfIntegrator::updated_blocks_ member (esdf_integrator.cc:54,80)
constexpr uint32_t kFlagEsdfPendOpen = 0x4000;      // holds voxels pushed to open_ by addNewRobotPosition (:84)
constexpr uint32_t kFlagFree = 0x8000;              // slot sits on the free list (belongs to no block)
constexpr uint32_t kFlagEsdfDirty = 0x10000;        // the ESDF block's voxels changed since the host mirror last took them
constexpr uint32_t kFlagEsdfUnsettled = 0x20000;    // the ESDF block's voxels were written from outside (vbx_blocks_upload): its next
                                                    // lower-phase run relaxes every voxel, not only the shell (k_esdf_tile)
                                                    // (VBX_UPDATE_DIRTY; the wavefront writes blocks the reference never flags)
// everything that says "this slot holds a block of the ESDF layer" (shared by every removal path)
constexpr uint32_t kEsdfBits = kFlagEsdfAlloc | (kFlagUpdMask << kFlagEsdfUpdShift) | kFlagEsdfPendClassify | kFlagEsdfPendOpen | kFlagEsdfDirty |
                               kFlagEsdfUnsettled;

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
  uint32_t esdf_phase_changed[2];  // raise / lower: the last update-wide sweep number in which a block changed
  uint32_t act_count[3];
#ifdef VBX_FOLD_STATS
  uint32_t dbg[16];
#endif
  uint32_t fold_giant_count;     // runs of >= kFoldGiant updates handed to fold_giant_runs
  uint32_t fold_long_count[16];  // long runs handed to fold_long_runs, one list per stripe (same-address atomics
                                 // serialise at ~90 per microsecond: a single counter cost the Simple fold 2 ms)
  uint32_t redo_count;       // rays whose voxel list must be rebuilt after slot assignment
  uint32_t fix_count;        // list entries in blocks that got their pool slot after the list pass (k_fast_fixup)
  uint32_t fix_overflow;     // ... more of them than the fix-up buffer holds: the queued rays are rebuilt instead
  uint32_t fast_idle_sweep;  // 0xFFFFFFFF - index of the first Fast sweep that found no open ray (0: none yet)
  uint32_t tomb_count;       // tombstones in the hash table (recycled blocks)
  // observed-set replay rounds run in batches without a host check in between (vbx_host_tsdf.hpp)
  uint32_t rp_n;             // probes of the round's ray range (device-side size of the round's sort)
  uint32_t rp_overflow;      // a round needed more probes than the batch was sized for: the rest of the batch idles
  uint32_t rp_changed_round; // 1 + index of the last round in which a probe count moved
  int32_t bbox_min[3];       // Merged: bounding box of the cloud's endpoint voxels (k_bbox_reduce)
  int32_t bbox_max[3];
  uint32_t bbox_wide;        // a point whose voxel index is not usable for the box: absolute keys this frame
  uint32_t live_slots;       // k_reclaim: slots that still hold a block of either layer
  uint32_t newlog_count;     // entries of the new-block log (k_collect_new) the host has not drained yet
  uint32_t newlog_overflow;  // ... entries that did not fit
  unsigned long long total_keys;
  unsigned long long voxels_touched[64];  // distinct voxels updated, striped by workgroup for the same reason
  unsigned long long rays_cast;
  unsigned long long num_kept;
};

// Host-visible copy of DevState in page-locked, device-mapped host memory.  A read-back is a
// tiny kernel that writes this struct and then its sequence number; the host spins on the
// number.  hipMemcpyAsync + hipStreamSynchronize costs ~35 us per read-back (copy engine launch
// + interrupt-driven wait), and a Fast frame needs five of them.
struct StateMirror {
  DevState st;
  uint32_t extra[3];
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
  unsigned long long* blk_first;  // 1 per slot: first-touch rank of a block published by the call in flight, ~0 otherwise
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
  uint32_t take_limit;           // Fast: points whose place in the taking order (ThreadSafeIndex) is at or beyond this are not taken
                                 // (max_integration_time_s, tsdf_integrator.cc:496-499); ~0u otherwise
};

// Merged: bundle keys are the endpoint voxel index RELATIVE to the cloud's bounding box, packed in as many
// bits as the box needs (a room is 7-8 bits per axis where the absolute index reserves 21), so the bundling
// sort runs two radix passes instead of six.  Order = (clearing, z, y, x), as with the absolute keys.
struct KeyFrame {
  int xmin, ymin, zmin;
  int bx, by, bz;  // bits per axis
};
__host__ __device__ inline int keyframe_bits(const KeyFrame& f) { return f.bx + f.by + f.bz + 1; }  // + clearing

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
// Returns the table position of the key (0xFFFFFFFF if the table is full).
__device__ inline uint32_t map_insert_key_pos(const MapDev& m, uint64_t key, uint32_t* new_list,
                                              DevState* st) {
  uint32_t h = mix_key(key) & m.hmask;
  for (uint32_t probes = 0; probes <= m.hmask; ++probes) {
    uint64_t k = __hip_atomic_load(&m.hkeys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == key) return h;
    if (k == kEmptyKey) {
      const unsigned long long old =
          atomicCAS((unsigned long long*)&m.hkeys[h], (unsigned long long)kEmptyKey,
                    (unsigned long long)key);
      if (old == kEmptyKey) {
        const uint32_t i = atomicAdd(&st->new_count, 1u);
        if (i < m.cap_blocks) new_list[i] = h; else atomicOr(&st->error, 1u);
        return h;
      }
      if (old == key) return h;
    }
    h = (h + 1) & m.hmask;
  }
  atomicOr(&st->error, 1u);
  return 0xFFFFFFFFu;
}
__device__ inline void map_insert_key(const MapDev& m, uint64_t key, uint32_t* new_list,
                                      DevState* st) {
  (void)map_insert_key_pos(m, key, new_list, st);
}

// Marks a block as part of the Layer and sets all Update bits (tsdf_integrator.cc:128).  The
// common case — block already published and flagged this frame — is a plain L2 read: only
// the first toucher pays for the atomic, so hundreds of thousands of rays crossing ~200
// blocks do not serialise on ~200 addresses.
//
// `rank` (the integrators' emit kernels): position of this touch in the reference's single-threaded order —
// (ray order << 24) | step along the ray, Merged's clearing pass above bit 62.  The smallest rank a NEW block sees
// is the moment allocateStorageAndGetVoxelPtr emplaces it in temp_block_map_ (tsdf_integrator.cc:107-121); the
// sequence of those moments decides the container's iteration order and with it the order in which
// updateLayerWithStoredBlocks inserts the blocks into the Layer (:137-147).  kFlagNewThisCall is set in the same
// atomic as kFlagPublished, so a block that shows Published without it was part of the Layer before this call.
// (A block that is published between the reading of `cur` and the atomic by a non-ranked publisher cannot happen: inside an
// integrate call only the emit kernel publishes before the fold runs.)
constexpr unsigned long long kNoRank = ~0ull;
// Every toucher of a block may call this: one plain L2 read in the common case (block published and flagged).
__device__ inline void publish_block(const MapDev& m, uint32_t slot, DevState* st) {
  const uint32_t want = kFlagPublished | kFlagUpdMask;
  const uint32_t cur = __hip_atomic_load(&m.blk_flags[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((cur & want) == want) return;
  const uint32_t old = atomicOr(&m.blk_flags[slot], want);
  if (!(old & kFlagPublished)) atomicAdd(&st->blocks_published, 1u);
}
// The integrators' emit kernels, once per (ray, block it enters): acts on blocks that were NOT part of the Layer when the call
// began — publishes them and keeps the smallest rank.  Blocks the Layer already held are left alone here: their Update
// bits are set by the fold, where the keys are sorted by voxel and exactly one thread per touched block does it
// (publish_block above) — thousands of rays re-flagging the same ~100 blocks from the emit kernel, every frame a consumer
// had cleared a bit, was a same-address atomic storm (k_fast_emit 18 -> 62 us whenever an ESDF update ran in between).
__device__ inline void publish_new_block_ranked(const MapDev& m, uint32_t slot, DevState* st, unsigned long long rank) {
  if (!m.blk_first) return;   // order tracking off (vbx_set_block_order_tracking: delta maps of the sharding): the fold publishes
  const uint32_t want = kFlagPublished | kFlagUpdMask | kFlagNewThisCall;
  const uint32_t cur = __hip_atomic_load(&m.blk_flags[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((cur & kFlagPublished) && !(cur & kFlagNewThisCall)) return;   // part of the Layer before this call
  if ((cur & want) != want) {
    const uint32_t old = atomicOr(&m.blk_flags[slot], want);
    if (!(old & kFlagPublished)) atomicAdd(&st->blocks_published, 1u);
  }
  if (__hip_atomic_load(&m.blk_first[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > rank) atomicMin(&m.blk_first[slot], rank);
}

__device__ inline void unpack_parent_bits(uint32_t s, int* x, int* y, int* z) {
  *x = (int)(int8_t)((s >> 8) & 0xFF);
  *y = (int)(int8_t)((s >> 16) & 0xFF);
  *z = (int)(int8_t)((s >> 24) & 0xFF);
}

}  // namespace

