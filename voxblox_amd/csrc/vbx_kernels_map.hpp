// vbx_kernels_map.hpp — map maintenance, layer (de)serialization, host-mirror packing and multi-GPU merge kernels
// Part of libvbx_hip.so's single translation unit (included by vbx_hip.hip, in order).

namespace {
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
__global__ void k_publish_state(const DevState* st, StateMirror* out, const uint32_t* extra0, const uint32_t* extra1,
                                const uint32_t* extra2, uint32_t seq) {
  const uint32_t* src = reinterpret_cast<const uint32_t*>(st);
  uint32_t* dst = reinterpret_cast<uint32_t*>(&out->st);
  for (uint32_t i = threadIdx.x; i < sizeof(DevState) / 4; i += blockDim.x) dst[i] = src[i];
  if (threadIdx.x == 0 && extra0) out->extra[0] = *extra0;
  if (threadIdx.x == 1 && extra1) out->extra[1] = *extra1;
  if (threadIdx.x == 2 && extra2) out->extra[2] = *extra2;
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
  for (int i = 0; i < 16; ++i) st->fold_long_count[i] = 0;
  st->fold_giant_count = 0;
#ifdef VBX_FOLD_STATS
  for (int i = 0; i < 16; ++i) st->dbg[i] = 0;
#endif
  st->fast_idle_sweep = 0;
  st->redo_count = 0;
  st->fix_count = 0;
  st->fix_overflow = 0;
  st->bbox_min[0] = st->bbox_min[1] = st->bbox_min[2] = 0x7FFFFFFF;
  st->bbox_max[0] = st->bbox_max[1] = st->bbox_max[2] = -0x7FFFFFFF;
  st->bbox_wide = 0;
  st->rp_n = 0;
  st->rp_overflow = 0;
  st->rp_changed_round = 0;
  st->total_keys = 0;
  for (int i = 0; i < 64; ++i) st->voxels_touched[i] = 0;
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
  if (m.blk_first) m.blk_first[slot] = kNoRank;
  __hip_atomic_store(&m.hvals[h], slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The blocks the call published (kFlagNewThisCall) -> the new-block log: {first-touch rank, packed BlockIndex, call
// number}; the flag and the rank go back to their idle values.  The host drains the log when somebody asks for the
// order (vbx_blocks_new_ordered, a reference-order ESDF update).
struct NewLogEntry {
  unsigned long long rank;
  unsigned long long key;
  unsigned long long seq;
};
__global__ void k_collect_new(MapDev m, uint32_t used, NewLogEntry* log, uint32_t log_cap, unsigned long long seq, DevState* st) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= used) return;
  const uint32_t f = m.blk_flags[slot];
  if (!(f & kFlagNewThisCall)) return;
  m.blk_flags[slot] = f & ~kFlagNewThisCall;
  const unsigned long long rank = m.blk_first[slot];
  m.blk_first[slot] = kNoRank;
  if (f & kFlagFree) return;
  const uint32_t i = atomicAdd(&st->newlog_count, 1u);
  if (i >= log_cap) {
    atomicAdd(&st->newlog_overflow, 1u);
    return;
  }
  log[i].rank = rank;
  log[i].key = pack_block_key(m.blk_idx[3 * slot], m.blk_idx[3 * slot + 1], m.blk_idx[3 * slot + 2]);
  log[i].seq = seq;
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
// AoS voxels -> SoA pool: the inverse of k_pack_tsdf_aos / k_pack_esdf_aos (vbx_blocks_upload).
__global__ void k_unpack_tsdf_aos(MapDev m, const uint32_t* __restrict__ slots, const uint32_t* __restrict__ in) {
  const uint32_t slot = slots[blockIdx.x];
  if (slot == kInvalidSlot) return;
  const uint32_t* a = in + (size_t)blockIdx.x * m.nvox * 3;
  uint32_t* d = reinterpret_cast<uint32_t*>(m.dist) + (size_t)slot * m.nvox;
  uint32_t* w = reinterpret_cast<uint32_t*>(m.weight) + (size_t)slot * m.nvox;
  uint32_t* c = m.rgba + (size_t)slot * m.nvox;
  for (uint32_t i = threadIdx.x; i < m.nvox * 3; i += blockDim.x) {  // coalesced word reads
    const uint32_t v = i / 3, k = i % 3;
    (k == 0 ? d : (k == 1 ? w : c))[v] = a[i];
  }
}
__global__ void k_unpack_esdf_aos(uint32_t nvox, float* edist, uint32_t* estate, const uint32_t* __restrict__ slots,
                                  const uint32_t* __restrict__ in) {
  const uint32_t slot = slots[blockIdx.x];
  if (slot == kInvalidSlot) return;
  const uint32_t* a = in + (size_t)blockIdx.x * nvox * 5;  // {float d; bool obs, hall, in_queue, fixed; int32 parent[3]}
  for (uint32_t v = threadIdx.x; v < nvox; v += blockDim.x) {
    const uint32_t fl = a[5 * v + 1];
    const uint32_t st = ((fl & 0xFFu) ? 1u : 0u) | ((fl & 0xFF00u) ? 2u : 0u) | ((fl & 0xFF0000u) ? 4u : 0u) |
                        ((fl & 0xFF000000u) ? 8u : 0u) | ((a[5 * v + 2] & 0xFFu) << 8) | ((a[5 * v + 3] & 0xFFu) << 16) |
                        ((a[5 * v + 4] & 0xFFu) << 24);
    edist[(size_t)slot * nvox + v] = __uint_as_float(a[5 * v]);
    estate[(size_t)slot * nvox + v] = st;
  }
}
// Replaces one layer's flag bits of the listed blocks (vbx_blocks_upload): `keep` masks what survives.
__global__ void k_replace_block_flags(MapDev m, const uint32_t* __restrict__ slots, uint32_t n, uint32_t keep,
                                      uint32_t base_bits, const uint8_t* __restrict__ upd, int upd_shift,
                                      const uint8_t* __restrict__ has_data) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || slots[i] == kInvalidSlot) return;
  uint32_t f = (m.blk_flags[slots[i]] & keep) | base_bits | (((uint32_t)upd[i] & kFlagUpdMask) << upd_shift);
  if (has_data && has_data[i]) f |= kFlagHasData;
  m.blk_flags[slots[i]] = f;
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
// One row per listed block: three planes of nvox 32-bit words — distance, weight, the colour's four bytes — i.e. the
// delta voxel itself (12 B, what mergeVoxelAIntoVoxelB reads of voxel A, voxel_utils.cc:10-22).  The weighted sums
// w*d and w*channel are formed by the OWNER (k_merge_sums) with the same float operations the sender used to run, so
// the merged map is bit for bit what the six-plane rows of rounds 1-4 gave at half the bytes.
constexpr uint32_t kRowPlanes = 3;
__global__ void k_export_sums(MapDev m, const uint32_t* __restrict__ slots, float* out) {
  const uint32_t b = blockIdx.x;
  const uint32_t slot = slots[b];
  float* o = out + (size_t)b * kRowPlanes * m.nvox;
  for (uint32_t v = threadIdx.x; v < m.nvox; v += blockDim.x) {
    float d = 0.f, w = 0.f;
    uint32_t c = 0u;
    if (slot != kInvalidSlot) {
      const uint32_t gid = slot * m.nvox + v;
      w = m.weight[gid];
      d = m.dist[gid];
      c = m.rgba[gid];
    }
    o[v] = d; o[m.nvox + v] = w; o[2 * m.nvox + v] = __uint_as_float(c);
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

// One workgroup per DISTINCT block: the block's rows of the staging buffer (row_start[b] .. row_start[b+1]
// of `rows`, ascending = the order the caller listed them: sender rank, then key) are summed in that
// order, then A = {d = Swd/Sw, w = Sw, colour = round(Swc/Sw)} is merged into the stored voxel
// (mergeVoxelAIntoVoxelB, voxel_utils.cc:10-22).
__global__ void k_merge_sums(MapDev m, const uint32_t* __restrict__ slots, const uint32_t* __restrict__ row_start,
                             const uint32_t* __restrict__ rows, const float* __restrict__ in, int apply_caps,
                             float trunc, float max_weight, DevState* st) {
  const uint32_t b = blockIdx.x;
  const uint32_t slot = slots[b];
  if (slot == kInvalidSlot) return;
  const uint32_t r0 = row_start[b], r1 = row_start[b + 1];
  bool any = false;
  for (uint32_t v = threadIdx.x; v < m.nvox; v += blockDim.x) {
    float acc[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (uint32_t q = r0; q < r1; ++q) {
      const float* a = in + (size_t)rows[q] * kRowPlanes * m.nvox;
      const float w = a[m.nvox + v];
      const uint32_t c = __float_as_uint(a[2 * m.nvox + v]);
      acc[0] += w * a[v];
      acc[1] += w;
      acc[2] += w * (float)(c & 0xFF);
      acc[3] += w * (float)((c >> 8) & 0xFF);
      acc[4] += w * (float)((c >> 16) & 0xFF);
      acc[5] += w * (float)((c >> 24) & 0xFF);
    }
    const float wA = acc[1];
    if (!(wA > 0.0f)) continue;
    any = true;
    const uint32_t gid = slot * m.nvox + v;
    const float dA = acc[0] / wA;
    uint32_t cA = 0;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      const float c = roundf(acc[2 + ch] / wA);
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
    m.blk_flags[slot] = (layer == VBX_LAYER_ESDF) ? (f & ~kEsdfBits) : (f & kEsdfBits);
  }
}

// Layer::removeBlock for a list of blocks (layer.h:160-165), one workgroup per listed block: slots[b] = the block's
// pool slot or kInvalidSlot (unordered_map::erase of a missing key is a no-op).  Same effect per block as
// k_remove_distant; a BlockIndex listed twice is harmless (the second workgroup writes the same zeros).
__global__ void k_remove_listed(MapDev m, float* edist, uint32_t* estate, int layer, const uint32_t* __restrict__ slots) {
  const uint32_t slot = slots[blockIdx.x];
  if (slot == kInvalidSlot) return;
  const uint32_t f = m.blk_flags[slot];
  const uint32_t need = (layer == VBX_LAYER_ESDF) ? kFlagEsdfAlloc : kFlagPublished;
  if (!(f & need)) return;
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
  if (threadIdx.x == 0) m.blk_flags[slot] = (layer == VBX_LAYER_ESDF) ? (f & ~kEsdfBits) : (f & kEsdfBits);
}

// Block::updated().reset(bits) on every block of one layer
__global__ void k_clear_update_bits(MapDev m, uint32_t n_slots, uint32_t need, uint32_t bits) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  const uint32_t f = m.blk_flags[s];
  if ((f & need) && (f & bits)) m.blk_flags[s] = f & ~bits;
}
// Slot recycling (Layer::removeBlock / removeDistantBlocks / removeAllBlocks free their blocks,
// layer.h:160-182).  A pool slot whose block belongs to neither layer any more — removed from both, or a
// candidate no ray ever reached — gives its hash entry up (tombstone) and goes on the free list;
// k_assign_slots hands free slots out before it grows the pool.  The block's voxels are zero at this
// point: the removal paths zero them, and a never-published candidate was never written.
__global__ void k_reclaim(MapDev m, DevState* st) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= st->pool_used) return;
  const uint32_t f = m.blk_flags[s];
  if (f & kFlagFree) return;
  if (f & (kFlagPublished | kFlagEsdfAlloc)) {
    atomicAdd(&st->live_slots, 1u);
    return;
  }
  const uint64_t key = pack_block_key(m.blk_idx[3 * s], m.blk_idx[3 * s + 1], m.blk_idx[3 * s + 2]);
  uint32_t h = mix_key(key) & m.hmask;
  for (uint32_t probes = 0; probes <= m.hmask; ++probes) {
    const uint64_t k = m.hkeys[h];
    if (k == key) {
      m.hkeys[h] = kTombKey;
      m.hvals[h] = kInvalidSlot;
      atomicAdd(&st->tomb_count, 1u);
      break;
    }
    if (k == kEmptyKey) break;  // (not reachable: every assigned slot has its entry)
    h = (h + 1) & m.hmask;
  }
  m.blk_flags[s] = kFlagFree;
  m.free_list[atomicAdd(&st->free_count, 1u)] = s;
}
__global__ void k_zero_free_slots_u32(MapDev m, uint32_t* per_voxel) {
  const uint32_t s = blockIdx.x;
  if (!(m.blk_flags[s] & kFlagFree)) return;
  for (uint32_t v = threadIdx.x; v < m.nvox; v += blockDim.x) per_voxel[(size_t)s * m.nvox + v] = 0u;
}
// Rebuilds the hash table from the pool (drops the tombstones): launched over an emptied table.
__global__ void k_rehash(MapDev m, DevState* st) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s == 0) st->tomb_count = 0;
  if (s >= st->pool_used) return;
  if (m.blk_flags[s] & kFlagFree) return;
  const uint64_t key = pack_block_key(m.blk_idx[3 * s], m.blk_idx[3 * s + 1], m.blk_idx[3 * s + 2]);
  uint32_t h = mix_key(key) & m.hmask;
  for (uint32_t probes = 0; probes <= m.hmask; ++probes) {
    const unsigned long long old = atomicCAS((unsigned long long*)&m.hkeys[h], (unsigned long long)kEmptyKey,
                                             (unsigned long long)key);
    if (old == kEmptyKey) {
      m.hvals[h] = s;
      return;
    }
    h = (h + 1) & m.hmask;
  }
}
// After an allocation pass that could not be served (pool at its limit): keys that were inserted but
// never got a slot leave the table again, so the next call sees a consistent map instead of entries that
// resolve to no block.
__global__ void k_drop_unassigned_keys(MapDev m, DevState* st) {
  const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h > m.hmask) return;
  const uint64_t k = m.hkeys[h];
  if (k == kEmptyKey || k == kTombKey) return;
  if (m.hvals[h] != kInvalidSlot) return;
  m.hkeys[h] = kTombKey;
  atomicAdd(&st->tomb_count, 1u);
}
// Every block is gone: back to the state of a new map (the voxel arrays are already zero).
__global__ void k_reset_pool(DevState* st) {
  st->pool_used = 0;
  st->free_count = 0;
  st->tomb_count = 0;
  st->new_count = 0;
}

__global__ void k_reset_tsdf_flags(MapDev m, uint32_t n_slots) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  m.blk_flags[s] &= (kFlagEsdfAlloc | (kFlagUpdMask << kFlagEsdfUpdShift));
}

}  // namespace

