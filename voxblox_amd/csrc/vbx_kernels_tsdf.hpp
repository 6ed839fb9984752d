// vbx_kernels_tsdf.hpp — ray tables, the generic ray march, the ordered per-voxel fold and Merged's bundling
// Part of libvbx_hip.so's single translation unit (included by vbx_hip.hip, in order).

namespace {
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
                              float* pcx, float* pcy, float* pcz, const uint32_t* __restrict__ s_of_p,
                              float voxel_size_inv, uint64_t* fast_keys, uint32_t* fast_vals, int32_t* bbox_partial,
                              DevState* st) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (bbox_partial) {
    // Merged: bounding box of the endpoint voxels (bundleRays' keys, tsdf_integrator.cc:359-361), one
    // partial result per workgroup; k_bbox_reduce folds them
    int lo[3] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF}, hi[3] = {-0x7FFFFFFF, -0x7FFFFFFF, -0x7FFFFFFF};
    if (p < n) {
      const f3 pc{pts[3 * p], pts[3 * p + 1], pts[3 * p + 2]};
      bool clearing = false;
      if (point_valid(c, pc, freespace != 0, &clearing)) {
        const f3 pg = pose_transform(T, pc);
        const float far = fabsf(pg.x) + fabsf(pg.y) + fabsf(pg.z);  // NaN / inf in any component propagates
        if (far * voxel_size_inv < 1048000.0f) {
          const l3 g = grid_index_from_point(pg, voxel_size_inv);
          lo[0] = hi[0] = (int)g.x; lo[1] = hi[1] = (int)g.y; lo[2] = hi[2] = (int)g.z;
        } else {
          // a non-finite point (the reference never sees one: conversions.h:135-137; the oracle lets it form a
          // bundle that casts nothing) or one beyond the key range: its voxel index is garbage and
          // must not stretch the box — the frame falls back to the absolute 3 x 21-bit keys, under which such a
          // point makes exactly the bundle it made before the keys became box-relative
          st->bbox_wide = 1;
        }
      }
    }
    __shared__ int s_lo[4][3], s_hi[4][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) {
        lo[a] = min(lo[a], __shfl_xor(lo[a], d));
        hi[a] = max(hi[a], __shfl_xor(hi[a], d));
      }
      if ((threadIdx.x & 63) == 0) {
        s_lo[threadIdx.x >> 6][a] = lo[a];
        s_hi[threadIdx.x >> 6][a] = hi[a];
      }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
      const int a = threadIdx.x;
      bbox_partial[6 * blockIdx.x + a] = min(min(s_lo[0][a], s_lo[1][a]), min(s_lo[2][a], s_lo[3][a]));
      bbox_partial[6 * blockIdx.x + 3 + a] = max(max(s_hi[0][a], s_hi[1][a]), max(s_hi[2][a], s_hi[3][a]));
    }
  }
  if (p >= n) return;
  const size_t s = s_of_p ? (size_t)s_of_p[p] : mixed_index_inverse(p, n);
  const f3 pc{pts[3 * p], pts[3 * p + 1], pts[3 * p + 2]};
  bool clearing = false;
  const bool valid = point_valid(c, pc, freespace != 0, &clearing) && s < (size_t)c.take_limit;
  const f3 pg = pose_transform(T, pc);
  tab.px[s] = pg.x;
  tab.py[s] = pg.y;
  tab.pz[s] = pg.z;
  tab.rgba[s] = rgba[p];
  tab.w[s] = voxel_weight(c, pc);
  tab.flags[s] = (valid ? 1 : 0) | (clearing ? 2 : 0);
  if (valid) {
    // voxel and block keys pack 21 bits per axis (pack_block_key and the voxel keys of the
    // bundle / emit kernels): a ray that leaves +-2^20 voxels fails the call instead of aliasing
    // (a clearing ray is cut to max_ray_length_m from the sensor, integrator_utils.cc:80-86: a far no-return
    // sentinel is as harmless as in the reference, what counts is where the sensor is)
    const float lim = 1048576.0f - (c.trunc + c.max_ray_length_m) * voxel_size_inv - 8.0f;
    // Merged bundles clearing points by their ENDPOINT voxel (tsdf_integrator.cc:352-367; pcx != nullptr marks
    // that caller): there the endpoint itself must fit the key, or distinct far endpoints would alias into one
    // bundle — such a cloud fails loudly instead.
    const f3 q = (clearing && !pcx) ? c.origin : pg;
    const float far = fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fabsf(q.z)) * voxel_size_inv;
    if (far >= lim && far < __builtin_inff()) {  // non-finite points are dropped further down
      atomicOr(&st->error, 8u);
      tab.flags[s] = 0;  // nothing of it is integrated; the call reports VBX_ERR_INVALID
    }
  }
  if (fast_keys) {
    // Fast: start_voxel_approx_set_.replaceHash(cell at start_voxel_subsampling_factor x resolution),
    // tsdf_integrator.cc:514-519.  key = slot << 32 | s so that a stable radix sort groups the probes of one
    // ApproxHashSet slot in visiting order; val = the 32-bit hash.
    uint64_t key = ~0ull;
    uint32_t h = 0;
    if (tab.flags[s] & 1) {
      h = long_index_hash(grid_index_from_point(pg, c.start_factor_times_inv));
      key = ((uint64_t)(h & 0xFFFFFu) << 32) | (uint64_t)s;
    }
    fast_keys[s] = key;
    fast_vals[s] = h;
  }
  if (pcx) {  // Merged keeps point_C for the bundle mean
    pcx[s] = pc.x;
    pcy[s] = pc.y;
    pcz[s] = pc.z;
  }
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
                            const uint32_t* __restrict__ limit, uint32_t* cnt, int want_totals, DevState* st) {
  const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  RayCaster rc;
  uint32_t n = 0;
  bool cast = false;
  if (o < tab.R && ray_init(rc, tab, o, c, m, from_origin != 0, nullptr)) {
    cast = true;
    n = (rc.cur == 0) ? rc.steps + 1 : 0;
    if (limit) n = min(n, limit[o]);
  }
  if (o <= tab.R) cnt[o] = n;  // cnt[R] = 0 terminates the scan
  if (!want_totals) return;    // (uniform: a kernel argument)
  // rays actually cast, and — where rays x longest path could pass 2^32 — a 64-bit total beside the 32-bit
  // offsets, so that such a cloud fails instead of wrapping.  One atomic per workgroup each: same-address
  // atomics serialise at ~90 per microsecond, a per-wave atomic made this kernel 10x slower.
  __shared__ unsigned long long s_sum[4];
  __shared__ uint32_t s_cast[4];
  unsigned long long sum = n;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
  const unsigned long long casts = __popcll(__ballot(cast));
  if ((threadIdx.x & 63) == 0) {
    s_sum[threadIdx.x >> 6] = sum;
    s_cast[threadIdx.x >> 6] = (uint32_t)casts;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
    const uint32_t cc = s_cast[0] + s_cast[1] + s_cast[2] + s_cast[3];
    if ((want_totals & 2) && t) atomicAdd(&st->total_keys, t);
    if ((want_totals & 1) && cc) atomicAdd(&st->rays_cast, (unsigned long long)cc);
  }
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
                           uint32_t n_graze, int two_pass, DevState* st) {
  const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= tab.R) return;
  RayCaster rc;
  if (!ray_init(rc, tab, o, c, m, from_origin != 0, nullptr)) return;
  if (rc.cur != 0) return;
  const uint32_t n = min(limit ? limit[o] : 0xFFFFFFFFu, rc.steps + 1);
  const bool clearing = (tab.flags[o] & 2) != 0;
  // Merged runs its clearing bundles as a second pass with its own updateLayerWithStoredBlocks (tsdf_integrator.cc:329-336)
  const unsigned long long rank_hi = ((unsigned long long)o << 24) | ((two_pass && clearing) ? 1ull << 62 : 0ull);
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
          publish_new_block_ranked(m, slot, st, rank_hi | (unsigned long long)(k & 0xFFFFFFu));
        }
      }
      if (slot != kInvalidSlot) out = ((uint64_t)(slot * m.nvox + bw.lin) << 32) | o;
    }
    keys[base + k] = out;
    bw.step(m.vps, m.vps_log2);
  }
}

// ---------------------------------------------------------------------------
// kernels: ordered per-voxel fold == updateTsdfVoxel (tsdf_integrator.cc:150-209) applied to
// each voxel's updates in ascending integration order, split into its state-independent half
// (tsdf_update_inputs) and the state-dependent rest (tsdf_update_state).
// ---------------------------------------------------------------------------
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
constexpr uint32_t kFoldGiant = 8192;  // and these to the workgroup-cooperative one (fold_giant_runs)
constexpr int kGiantWaves = 16;
constexpr int kGU = 8;  // chunks of 64 updates per wave and round
constexpr uint32_t kGiantCap = 1u << 16;  // giant runs listed per call; beyond that they are folded as long runs

__device__ inline l3 voxel_of_gid(const MapDev& m, uint32_t gid) {
  const uint32_t slot = gid / m.nvox;
  const uint32_t lin = gid - slot * m.nvox;
  const int lx = lin & (m.vps - 1);
  const int ly = (lin >> m.vps_log2) & (m.vps - 1);
  const int lz = lin >> (2 * m.vps_log2);
  return {(long long)m.blk_idx[3 * slot] * m.vps + lx, (long long)m.blk_idx[3 * slot + 1] * m.vps + ly,
          (long long)m.blk_idx[3 * slot + 2] * m.vps + lz};
}

// One lane's update applied literally to (d, Wpre, col): does it give exactly (d, Wpost, col)?
__device__ inline bool update_keeps(const CastCfg& c, float sdf, float uw, uint32_t color, float d, float Wpre,
                                    float Wpost, uint32_t col) {
  float d1 = d, W1 = Wpre;
  uint32_t c1 = col;
  tsdf_update_state(c, sdf, uw, color, d1, W1, c1);
  return __float_as_uint(d1) == __float_as_uint(d) && __float_as_uint(W1) == __float_as_uint(Wpost) && c1 == col;
}

// The state-independent half of every update (sdf and adjusted weight, colour) in sorted key order:
// the gathers by ray index run fully parallel here, and the ordered folds below read consecutive
// memory instead of chasing a ray index per step (k_fold spent ~1 us per update on that chain).
// ident (optional): one byte per aligned segment of 256 keys = per workgroup — 1 when the whole segment
// belongs to one voxel and every update in it leaves that voxel's CURRENT state untouched (a
// saturated free-space voxel next to the sensor).  fold_giant_runs jumps over such segments while the
// voxel still has the state it had here.
__global__ void __launch_bounds__(256) k_fold_inputs(const uint64_t* __restrict__ keys, size_t n, RayTab tab, CastCfg c,
                                                     MapDev m, float* in_sdf, float* in_uw, uint32_t* in_col,
                                                     uint8_t* ident) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t key = (i < n) ? keys[i] : ~0ull;
  const bool live = key != ~0ull;
  const uint32_t gid = (uint32_t)(key >> 32);
  float sdf = 0.f, uw = 0.f;
  uint32_t color = 0;
  if (live) {
    const uint32_t o = (uint32_t)(key & 0xFFFFFFFFu);
    const f3 pg{tab.px[o], tab.py[o], tab.pz[o]};
    tsdf_update_inputs(c, m.voxel_size, pg, voxel_of_gid(m, gid), tab.w[o], &sdf, &uw);
    color = tab.rgba[o];
    in_sdf[i] = sdf;
    in_uw[i] = uw;
    in_col[i] = color;
  }
  if (!ident) return;
  __shared__ uint32_t s_gid;
  if (threadIdx.x == 0) s_gid = gid;
  __syncthreads();
  const bool one_voxel = __syncthreads_and(live && gid == s_gid) != 0;
  bool keeps = false;
  if (one_voxel) {
    const float d0 = m.dist[gid], W0 = m.weight[gid];
    keeps = update_keeps(c, sdf, uw, color, d0, W0, W0, m.rgba[gid]);
  }
  const bool all_keep = __syncthreads_and(keeps) != 0;
  if (threadIdx.x == 0) ident[blockIdx.x] = all_keep ? 1 : 0;
}

// The whole fold in one kernel for the Fast integrator, whose runs are two or three updates (a voxel is updated
// again only when the approximate observed set forgot it): one thread per key, a run's head
// computes the inputs of its updates on the spot and applies them in order.  Two launches less per
// frame than inputs / tiles / long runs, and no staging arrays.
__global__ void __launch_bounds__(256) k_fold_direct(const uint64_t* __restrict__ keys, size_t n, RayTab tab, CastCfg c,
                                                     MapDev m, DevState* st) {
  // Round 6: the state-independent half of EVERY update (projective sdf, drop-off weight: the gathers from the ray table, a
  // square root and two divisions) is computed by the thread that holds the key — all lanes busy — and left in LDS; the run's
  // head then only applies the two or three cheap state updates of its voxel.  Before, the head walked its run alone through
  // key -> ray table -> arithmetic, one dependent chain per update, while two thirds of the lanes had left (64 us per frame).
  __shared__ float s_sdf[256], s_uw[256];
  __shared__ uint32_t s_rgba[256], s_gid[256];
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t key = (i < n) ? keys[i] : ~0ull;
  const uint32_t gid = (uint32_t)(key >> 32);
  const uint32_t prev_gid = i > 0 ? (uint32_t)(keys[i - 1] >> 32) : 0xFFFFFFFFu;
  const bool head = key != ~0ull && !(i > 0 && prev_gid == gid);
  l3 g{0, 0, 0};
  s_gid[threadIdx.x] = gid;
  if (key != ~0ull) {
    g = voxel_of_gid(m, gid);
    const uint32_t o = (uint32_t)(key & 0xFFFFFFFFu);
    float sdf, uw;
    tsdf_update_inputs(c, m.voxel_size, f3{tab.px[o], tab.py[o], tab.pz[o]}, g, tab.w[o], &sdf, &uw);
    s_sdf[threadIdx.x] = sdf;
    s_uw[threadIdx.x] = uw;
    s_rgba[threadIdx.x] = tab.rgba[o];
  }
  const int nheads = __syncthreads_count(head);
  if (threadIdx.x == 0 && nheads) atomicAdd(&st->voxels_touched[blockIdx.x & 63u], (unsigned long long)nheads);
  if (!head) return;
  // block->updated().set() (tsdf_integrator.cc:128): the keys are sorted by voxel, so the first key of a block is one thread
  if (i == 0 || prev_gid / m.nvox != gid / m.nvox) publish_block(m, gid / m.nvox, st);
  float d = m.dist[gid];
  float W = m.weight[gid];
  uint32_t col = m.rgba[gid];
  uint32_t t = threadIdx.x;
  for (size_t j = i;;) {
    if (t < 256u) {   // the update's inputs wait in LDS
      tsdf_update_state(c, s_sdf[t], s_uw[t], s_rgba[t], d, W, col);
    } else {          // the run crosses the workgroup's last key: the old way for the rest of it
      const uint32_t o = (uint32_t)(keys[j] & 0xFFFFFFFFu);
      float sdf, uw;
      tsdf_update_inputs(c, m.voxel_size, f3{tab.px[o], tab.py[o], tab.pz[o]}, g, tab.w[o], &sdf, &uw);
      tsdf_update_state(c, sdf, uw, tab.rgba[o], d, W, col);
    }
    ++t;
    if (++j >= n) break;
    if (t < 256u ? s_gid[t] != gid : (uint32_t)(keys[j] >> 32) != gid) break;
  }
  m.dist[gid] = d;
  m.weight[gid] = W;
  m.rgba[gid] = col;
}

// The fold proper.  Every workgroup owns a tile of 4096 consecutive keys: it finds the heads of the runs
// (the keys of a voxel are contiguous after the sort), hands long runs to fold_long_runs and giant ones to
// fold_giant_runs, both in k_fold_runs (one look ahead at distance kFoldShort / kFoldGiant tells), collects the short ones in
// LDS and then folds those one run per thread.  Collecting first matters: with a hundred updates per
// voxel only one key in a hundred is a head, and a head walking its run in place kept a whole wave
// resident for one or two active lanes (0.9 ms of dependent loads per Simple frame).
template <int kFoldTile>
__global__ void __launch_bounds__(256) k_fold(const uint64_t* __restrict__ keys, size_t n, CastCfg c, MapDev m,
                                              const float* __restrict__ in_sdf, const float* __restrict__ in_uw,
                                              const uint32_t* __restrict__ in_col, uint32_t* long_list,
                                              uint32_t long_cap, uint32_t* giant_list, DevState* st) {
  __shared__ uint32_t s_short[256 * kFoldTile];
  __shared__ uint32_t s_long[256 * kFoldTile / kFoldShort + 8];  // long heads are more than kFoldShort apart
  __shared__ uint32_t s_nshort, s_nlong, s_nheads, s_base;
  if (threadIdx.x == 0) s_nshort = s_nlong = s_nheads = 0;
  __syncthreads();
  const size_t tile0 = (size_t)blockIdx.x * (256 * kFoldTile);
  uint32_t nheads = 0;
#pragma unroll 4
  for (int t = 0; t < kFoldTile; ++t) {
    const size_t i = tile0 + (size_t)t * 256 + threadIdx.x;
    if (i >= n) break;
    const uint64_t key = keys[i];
    const uint32_t gid = (uint32_t)(key >> 32);
    const uint32_t prev_gid = i > 0 ? (uint32_t)(keys[i - 1] >> 32) : 0xFFFFFFFFu;
    if (key == ~0ull || (i > 0 && prev_gid == gid)) continue;  // not a head
    ++nheads;
    // block->updated().set() (tsdf_integrator.cc:128): the first key of a block in sorted order is one thread
    if (i == 0 || prev_gid / m.nvox != gid / m.nvox) publish_block(m, gid / m.nvox, st);
    if (i + kFoldShort < n && (uint32_t)(keys[i + kFoldShort] >> 32) == gid) {
      if (giant_list && i + kFoldGiant < n && (uint32_t)(keys[i + kFoldGiant] >> 32) == gid) {
        const uint32_t idx = atomicAdd(&st->fold_giant_count, 1u);  // a hundred per frame
        if (idx < kGiantCap) {
          giant_list[idx] = (uint32_t)i;
          continue;
        }
      }
      s_long[atomicAdd(&s_nlong, 1u)] = (uint32_t)i;
    } else {
      s_short[atomicAdd(&s_nshort, 1u)] = (uint32_t)i;
    }
  }
  if (nheads) atomicAdd(&s_nheads, nheads);
  __syncthreads();
  // same-address atomics retire at ~90 per microsecond: one per workgroup and counter, counters striped
  const uint32_t stripe = blockIdx.x & 15u;
  if (threadIdx.x == 0) {
    if (s_nheads) atomicAdd(&st->voxels_touched[blockIdx.x & 63u], (unsigned long long)s_nheads);
    if (s_nlong) s_base = atomicAdd(&st->fold_long_count[stripe], s_nlong);
  }
  __syncthreads();
  if (threadIdx.x < s_nlong) long_list[(size_t)stripe * long_cap + s_base + threadIdx.x] = s_long[threadIdx.x];

  for (uint32_t e = threadIdx.x; e < s_nshort; e += 256) {
    size_t j = s_short[e];
    const uint32_t gid = (uint32_t)(keys[j] >> 32);
    float d = m.dist[gid];
    float W = m.weight[gid];
    uint32_t col = m.rgba[gid];
    bool more = true;
    while (more) {  // four updates per trip: their loads do not depend on each other
      bool mine[4];
      float sdf[4], uw[4];
      uint32_t color[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) mine[u] = j + u < n && (uint32_t)(keys[j + u] >> 32) == gid;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        mine[u] = mine[u] && (u == 0 || mine[u - 1]);
        sdf[u] = mine[u] ? in_sdf[j + u] : 0.f;
        uw[u] = mine[u] ? in_uw[j + u] : 0.f;
        color[u] = mine[u] ? in_col[j + u] : 0u;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (mine[u]) tsdf_update_state(c, sdf[u], uw[u], color[u], d, W, col);
      more = mine[3];
      j += 4;
    }
    m.dist[gid] = d;
    m.weight[gid] = W;
    m.rgba[gid] = col;
  }
}

// Long runs (the voxels around the sensor origin collect one update per ray) are folded by whole
// waves, 64 updates per step.  The state-independent part of the 64 updates (sdf, weight) was
// computed in parallel by k_fold_inputs; the ordered fold over them is done by the cheapest exact
// method:
//   1. the distance stays where it is (clamped at +-trunc) and no colour changes — free space, the
//      bulk of a frame: only the weight chain W <- min(max_weight, W + w) is evaluated in order
//      (weight_stretches), then every update is verified in parallel: applied literally to
//      (d, the weight in front of it, col) it must give (d, the weight behind it, col) — by
//      induction over the lanes that IS the sequential result (and for a saturated voxel the
//      identity);
//   2. otherwise the 64 updates are applied in order — the weights still come from the chain, the
//      state-independent operands are precomputed per lane and broadcast.
// Both produce exactly the sequential result of updateTsdfVoxel.

// The weight of one binade as integers: W = k0 * 2^(e-23).
struct Binade {
  int e, k0;
};
__device__ inline Binade binade_of(float W) {
  const uint32_t wb = __float_as_uint(W);
  return {(int)((wb >> 23) & 0xFFu) - 127, (int)(wb & 0x7FFFFFu) | 0x800000};
}
// rn(w / u) for u = 2^(e-23), or `exact = false` when fl(W + w) cannot be written as (k + that) * u for
// every k of the binade: negative or huge w, or w / u exactly halfway (ties depend on k).
__device__ inline int binade_increment(float uw, int e, bool* exact) {
  const float f = ldexpf(uw, 23 - e);
  *exact = uw >= 0.0f && f < 8388608.0f && (f - floorf(f)) != 0.5f;
  return *exact ? (int)rintf(f) : 0;
}
// Inclusive prefix sum over the 64 lanes on the DPP data path (no LDS round trips): shifts inside the
// rows of 16, then the row totals are broadcast into the following rows.
__device__ inline int wave_prefix_incl(int v, int /*lane*/) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return v;
}

// The chain W <- min(max_weight, W + w_j) over the first `cnt` lanes, in lane order; returns the final
// W and leaves in *Wmine the value each lane's update starts from.  Float addition does not associate,
// but the chain need not run one element at a time: while W stays inside one binade [2^e, 2^(e+1)) it
// is an integer multiple k of u = 2^(e-23), and
//     fl(W + w) = (k + rn(w / u)) * u        (w >= 0, w / u not a tie, k + w / u < 2^24),
// so the chain over the lanes is an INTEGER prefix sum of rn(w_j / u), which associates.  Lanes are
// consumed in stretches: a stretch ends where the binade is left, max_weight is reached, or a tie /
// negative weight / W < epsilon needs the literal rule; that one element is then applied as written in
// updateTsdfVoxel (tsdf_integrator.cc:183-195) and the next stretch starts behind it.
// (tests/cpp/weight_chain_model.c checks the same procedure against the plain loop on the CPU; on the
// device every result is verified by the caller anyway.)
__device__ inline float weight_stretches(const CastCfg& c, float W, float uw, int cnt, int lane, float* Wmine_out,
                                         float* Wpost_out) {
  float Wrun = W, Wmine = W, Wpost = W;  // per lane: the weight in front of and behind its update
  int start = 0;
  while (start < cnt) {
    if (Wrun == c.max_weight && __all(uw >= 0.0f)) {  // saturated: min(max, max + w) == max
      if (lane >= start) Wmine = Wpost = Wrun;
      break;
    }
    const Binade bn = binade_of(Wrun);
    const bool active = lane >= start && lane < cnt;
    bool exact = false;
    int inc = 0;
    if (active && Wrun >= 1e-6f) inc = binade_increment(uw, bn.e, &exact);
    const int pre = wave_prefix_incl(inc, lane);
    const float Wafter = ldexpf((float)(bn.k0 + pre), bn.e - 23);  // exact where k0 + pre < 2^24
    const bool good = !active || (exact && (bn.k0 + pre) <= 0xFFFFFF && Wafter <= c.max_weight);
    const unsigned long long bad = __ballot(!good);
    const int stop = bad ? (__ffsll((long long)bad) - 1) : 64;  // first lane outside the stretch
    if (stop > start) {
      if (lane >= start && lane < stop) {
        Wmine = ldexpf((float)(bn.k0 + pre - inc), bn.e - 23);
        Wpost = Wafter;
      }
      Wrun = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Wafter), stop - 1));
      start = stop;
    } else {  // element `start` by the literal rule
      const float uws = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(uw), start));
      const float nw = Wrun + uws;
      const float Wnew = (nw < 1e-6f) ? Wrun : std_min(c.max_weight, nw);
      if (lane == start) {
        Wmine = Wrun;
        Wpost = Wnew;
      }
      Wrun = Wnew;
      ++start;
    }
  }
  *Wmine_out = Wmine;
  *Wpost_out = Wpost;
  return Wrun;
}

// Folds the first `cnt` lanes' updates, in lane order, into (d, W, col).  Wave-uniform state.
__device__ inline void fold_chunk(const CastCfg& c, int cnt, float sdf, float uw, uint32_t color, int lane, float& d,
                                  float& W, uint32_t& col, DevState* st) {
  // 1. weight chain, then every update verified against its own weights (a saturated voxel falls out
  // of weight_stretches at once and this is the identity test)
  float Wmine, Wpost;
  const float Wend = weight_stretches(c, W, uw, cnt, lane, &Wmine, &Wpost);
  {
    const bool all_ok = __all(lane >= cnt || update_keeps(c, sdf, uw, color, d, Wmine, Wpost, col));
#ifdef VBX_FOLD_STATS
    if (lane == 0) atomicAdd(&st->act_count[all_ok ? 0 : 1], 1u);
#endif
    if (all_ok) {
      W = Wend;
      return;
    }
  }
  // 2. ordered application.  The weights depend on neither d nor col, so the chain above stands; what
  // is left in order are two short dependent chains (d, and col where |sdf| < trunc) whose other
  // operands — products, sums, the normalised blend weights with their two divisions — are computed
  // for all 64 updates at once.  (The generic loop below cost ~100 instructions per update, executed by
  // a whole wave for one voxel: 7 % of the chunks took half of the kernel.)
#ifdef VBX_FOLD_STATS
  if (lane == 0) atomicAdd(&st->act_count[2], 1u);
#endif
  const float nw = Wmine + uw;
  const bool skipped = nw < 1e-6f;  // updateTsdfVoxel returns before touching anything (tsdf_integrator.cc:192-194)
  const float Wlit = skipped ? Wmine : std_min(c.max_weight, nw);
  if (__all(lane >= cnt || __float_as_uint(Wlit) == __float_as_uint(Wpost))) {  // the chain is the literal one
    const unsigned long long act = __ballot(lane < cnt && !skipped);
    const unsigned long long inb = __ballot(lane < cnt && fabsf(sdf) < c.trunc);
    const float p = sdf * uw;
    const float w1n = Wmine / nw, w2n = uw / nw;  // Color::blendTwoColors normalises by w1 + w2 == nw
    float cb[4];
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) cb[ch] = (float)(int)((color >> (8 * ch)) & 0xFF) * w2n;
#pragma unroll 1
    for (int j = 0; j < cnt; ++j) {
      if (!((act >> j) & 1ull)) continue;
      const float pj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), j));
      const float Wj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Wmine), j));
      const float nwj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nw), j));
      const float nsdf = (pj + d * Wj) / nwj;
      if ((inb >> j) & 1ull) {
        const float w1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w1n), j));
        uint32_t out = 0;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
          const float a = (float)(int)((col >> (8 * ch)) & 0xFF);
          const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cb[ch]), j));
          out |= ((uint32_t)(int)roundf(a * w1 + b) & 0xFFu) << (8 * ch);
        }
        col = out;
      }
      d = (nsdf > 0.0f) ? std_min(c.trunc, nsdf) : std_max(-c.trunc, nsdf);
    }
    W = Wend;
    return;
  }
  // 3. the literal loop (not reached as long as weight_stretches does what it says)
#pragma unroll 1
  for (int j = 0; j < cnt; ++j) {
    const float sj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sdf), j));
    const float wj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(uw), j));
    const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)color, j);
    tsdf_update_state(c, sj, wj, cj, d, W, col);
  }
}

// kU chunks of 64 updates of the run of voxel `gid`, starting at `base`: inputs are fetched without
// waiting for the keys (every index below n is readable), cnt[u] = how many lanes of chunk u still
// belong to the run (the keys of a voxel are contiguous, so they are a prefix).
template <int kU>
__device__ inline void load_chunks(const uint64_t* __restrict__ keys, size_t n, const float* __restrict__ in_sdf,
                                   const float* __restrict__ in_uw, const uint32_t* __restrict__ in_col, uint32_t gid,
                                   size_t base, int lane, float* sdf_u, float* uw_u, uint32_t* color_u, int* cnt_u) {
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    const size_t i = base + 64 * u + lane;
    const bool in = i < n;
    const uint64_t key = in ? keys[i] : ~0ull;
    const float s = in ? in_sdf[i] : 0.f;
    const float w = in ? in_uw[i] : 0.f;
    const uint32_t cc = in ? in_col[i] : 0u;
    const bool mine = (key != ~0ull) && ((uint32_t)(key >> 32) == gid);
    const unsigned long long V = __ballot(mine);
    cnt_u[u] = (V == ~0ull) ? 64 : (__ffsll((long long)~V) - 1);
    const bool live = lane < cnt_u[u];
    sdf_u[u] = live ? s : 0.f;
    uw_u[u] = live ? w : 0.f;
    color_u[u] = live ? cc : 0u;
  }
}

__device__ inline void fold_long_runs(uint32_t first_wave, uint32_t n_waves, const uint64_t* __restrict__ keys, size_t n,
                                      const CastCfg& c, const MapDev& m, const float* __restrict__ in_sdf,
                                      const float* __restrict__ in_uw, const uint32_t* __restrict__ in_col,
                                      const uint32_t* __restrict__ long_list, uint32_t long_cap, DevState* st) {
  const int lane = threadIdx.x & 63;
  uint32_t n_long = 0, per_stripe[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    per_stripe[q] = st->fold_long_count[q];
    n_long += per_stripe[q];
  }
#ifdef VBX_FOLD_STATS
  uint32_t dbg_iters = 0;
#endif
  for (uint32_t seg = first_wave; seg < n_long; seg += n_waves) {
    // seg-th entry over the 16 striped lists
    uint32_t rest = seg;
    int q = 0;
#pragma unroll
    for (int t = 0; t < 15; ++t)
      if (q == t && rest >= per_stripe[t]) {
        rest -= per_stripe[t];
        q = t + 1;
      }
    const size_t i0 = long_list[(size_t)q * long_cap + rest];
    const uint32_t gid = (uint32_t)(keys[i0] >> 32);
    float d = m.dist[gid];
    float W = m.weight[gid];
    uint32_t col = m.rgba[gid];
    constexpr int kU = 4;
    bool more = true;
#ifdef VBX_FOLD_STATS
    if (lane == 0) atomicAdd(&st->dbg[7], 1u);
#endif
    for (size_t base = i0; more; base += 64 * kU) {
      float sdf_u[kU], uw_u[kU];
      uint32_t color_u[kU];
      int cnt_u[kU];
      load_chunks<kU>(keys, n, in_sdf, in_uw, in_col, gid, base, lane, sdf_u, uw_u, color_u, cnt_u);
#ifdef VBX_FOLD_STATS
      ++dbg_iters;
      if (lane == 0) atomicAdd(&st->dbg[8], 1u);
#endif
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if (cnt_u[u] == 0) { more = false; break; }
        fold_chunk(c, cnt_u[u], sdf_u[u], uw_u[u], color_u[u], lane, d, W, col, st);
        if (cnt_u[u] < 64) { more = false; break; }
      }
    }
    if (lane == 0) {
      m.dist[gid] = d;
      m.weight[gid] = W;
      m.rgba[gid] = col;
    }
  }
#ifdef VBX_FOLD_STATS
  if (lane == 0) atomicMax(&st->dbg[9], dbg_iters);
#endif
}

// Giant runs (kFoldGiant updates and more: the sensor's own voxel collects one update per ray, its
// neighbours tens of thousands): a single wave walking 300k updates is a 4 ms chain of dependent
// loads, so these get a whole workgroup of 16 waves and proceed in rounds of 16 x kGU x 64 updates.
// In a round every wave takes one segment and assumes the cheap case for everything before it — the
// distance and colour stay put and the weight advances inside its current binade (or sits at
// max_weight) — under which the weight in front of every update is an integer prefix sum across the
// whole round (lanes, chunks, waves).  Each update is then applied literally to (d, its claimed W,
// col) and must return (d, the next update's claimed W, col); the prefix of segments for which that
// holds is exactly the sequential result (induction as in fold_chunk).  The first segment where it
// does not hold is folded by its wave with fold_chunk from the exact state in front of it, and the
// next round starts behind it.
__device__ inline void fold_giant_runs(uint32_t first, uint32_t stride, const uint64_t* __restrict__ keys, size_t n,
                                       const CastCfg& c, const MapDev& m, const float* __restrict__ in_sdf,
                                       const float* __restrict__ in_uw, const uint32_t* __restrict__ in_col,
                                       const uint32_t* __restrict__ giant_list, const uint8_t* __restrict__ ident,
                                       DevState* st) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  __shared__ float s_d, s_W;
  __shared__ uint32_t s_col;
  __shared__ int s_tot[kGiantWaves], s_act[kGiantWaves], s_ok[kGiantWaves], s_first[kGiantWaves];
  const uint32_t n_giant = min(st->fold_giant_count, kGiantCap);
  for (uint32_t g = first; g < n_giant; g += stride) {
    const size_t i0 = giant_list[g];
    const uint32_t gid = (uint32_t)(keys[i0] >> 32);
    __syncthreads();  // the previous run's last reads of the shared state
#ifdef VBX_FOLD_STATS
    uint32_t dbg_rounds = 0;
    if (threadIdx.x == 0) atomicAdd(&st->dbg[0], 1u);
#endif
    // the state the ident bytes were computed against
    const float d0 = m.dist[gid], W0 = m.weight[gid];
    const uint32_t col0 = m.rgba[gid];
    if (threadIdx.x == 0) {
      s_d = d0;
      s_W = W0;
      s_col = col0;
    }
    // rounds work on aligned segments of 256 keys (the granularity of ident): the run's head up to the
    // first boundary is folded by wave 0
    size_t pos = (i0 + 255) & ~(size_t)255;
    if (pos > n) pos = n;
    if (wave == 0 && pos > i0) {
      float sdf_u[kGU], uw_u[kGU];
      uint32_t color_u[kGU];
      int cnt_u[kGU];
      load_chunks<kGU>(keys, pos, in_sdf, in_uw, in_col, gid, i0, lane, sdf_u, uw_u, color_u, cnt_u);
      float d2 = d0, W2 = W0;
      uint32_t col2 = col0;
#pragma unroll
      for (int u = 0; u < kGU; ++u)
        if (cnt_u[u] > 0) fold_chunk(c, cnt_u[u], sdf_u[u], uw_u[u], color_u[u], lane, d2, W2, col2, st);
      if (lane == 0) {
        s_d = d2;
        s_W = W2;
        s_col = col2;
      }
    }
    while (true) {
      __syncthreads();
      const float d = s_d, W = s_W;
      const uint32_t col = s_col;
      if (__float_as_uint(d) == __float_as_uint(d0) && __float_as_uint(W) == __float_as_uint(W0) && col == col0) {
        // untouched so far: jump over the segments k_fold_inputs found to be identities, 1024 at a time
        const size_t seg = pos / 256 + threadIdx.x;
        const bool skip = (seg + 1) * 256 <= n && ident[seg] && (uint32_t)(keys[seg * 256 + 255] >> 32) == gid;
        const unsigned long long stay = __ballot(!skip);
        if (lane == 0) s_first[wave] = stay ? (__ffsll((long long)stay) - 1) : 64;
        __syncthreads();
        int first = 64 * kGiantWaves;
        for (int w2 = kGiantWaves - 1; w2 >= 0; --w2)
          if (s_first[w2] < 64) first = 64 * w2 + s_first[w2];
        pos += (size_t)first * 256;
#ifdef VBX_FOLD_STATS
        if (threadIdx.x == 0) { atomicAdd(&st->dbg[3], 1u); atomicAdd(&st->dbg[4], (uint32_t)first); }
#endif
        if (first == 64 * kGiantWaves) continue;
      }
      float sdf_u[kGU], uw_u[kGU];
      uint32_t color_u[kGU];
      int cnt_u[kGU], inc_u[kGU], pre_u[kGU];  // pre: inclusive prefix inside the segment
      load_chunks<kGU>(keys, n, in_sdf, in_uw, in_col, gid, pos + (size_t)wave * (64 * kGU), lane, sdf_u, uw_u,
                       color_u, cnt_u);
      const bool saturated = W == c.max_weight;
      const Binade bn = binade_of(W);
      bool claim = W >= 1e-6f;  // every update of the segment can be written as an integer step
      int run = 0, active = 0;
#pragma unroll
      for (int u = 0; u < kGU; ++u) {
        bool exact = true;
        int inc = 0;
        if (lane < cnt_u[u]) {
          if (saturated) exact = uw_u[u] >= 0.0f;
          else inc = binade_increment(uw_u[u], bn.e, &exact);
        }
        claim = claim && exact && inc <= 0xFFFF;  // keeps every sum of a round inside an int
        inc_u[u] = inc;
        pre_u[u] = run + wave_prefix_incl(inc, lane);
        run = __builtin_amdgcn_readlane(pre_u[u], 63);
        active += cnt_u[u];
      }
      if (lane == 0) {
        s_tot[wave] = run;
        s_act[wave] = active;
      }
      __syncthreads();
      int before = 0;
      for (int w2 = 0; w2 < wave; ++w2) before += s_tot[w2];
      // verify every update of the segment against its claimed weights
      bool ok = __all(claim) && (long long)bn.k0 + before + run <= 0xFFFFFF;
      if (ok) {
#pragma unroll
        for (int u = 0; u < kGU; ++u) {
          if (lane < cnt_u[u]) {
            const float Wpre = saturated ? W : ldexpf((float)(bn.k0 + before + pre_u[u] - inc_u[u]), bn.e - 23);
            const float Wpost = saturated ? W : ldexpf((float)(bn.k0 + before + pre_u[u]), bn.e - 23);
            ok = ok && update_keeps(c, sdf_u[u], uw_u[u], color_u[u], d, Wpre, Wpost, col);
          }
        }
        ok = __all(ok);
      }
      if (lane == 0) s_ok[wave] = ok ? 1 : 0;
      __syncthreads();
      // first segment that failed, first segment in which the run ended
      int fail = kGiantWaves, last = kGiantWaves, upto = 0;
      for (int w2 = 0; w2 < kGiantWaves; ++w2) {
        if (fail == kGiantWaves && !s_ok[w2]) fail = w2;
        if (last == kGiantWaves && s_act[w2] < 64 * kGU) last = w2;
      }
      const int good = min(fail, last == kGiantWaves ? kGiantWaves : last + 1);  // segments taken as claimed
      for (int w2 = 0; w2 < good; ++w2) upto += s_tot[w2];
      const float Wgood = saturated ? W : ldexpf((float)(bn.k0 + upto), bn.e - 23);
      const bool ended = last < fail || (fail == last && fail < kGiantWaves);  // the run ends inside this round
      __syncthreads();  // everyone has read s_d / s_W / s_ok
#ifdef VBX_FOLD_STATS
      ++dbg_rounds;
      if (threadIdx.x == 0) { atomicAdd(&st->dbg[1], 1u); if (fail < kGiantWaves && fail <= last) atomicAdd(&st->dbg[2], 1u); }
#endif
      if (fail < kGiantWaves && fail <= last) {
        if (wave == fail) {  // fold the failed segment from the exact state in front of it
          float d2 = d, W2 = Wgood;
          uint32_t col2 = col;
#pragma unroll
          for (int u = 0; u < kGU; ++u)
            if (cnt_u[u] > 0) fold_chunk(c, cnt_u[u], sdf_u[u], uw_u[u], color_u[u], lane, d2, W2, col2, st);
          if (lane == 0) {
            s_d = d2;
            s_W = W2;
            s_col = col2;
          }
        }
        pos += (size_t)(fail + 1) * (64 * kGU);
      } else {
        if (threadIdx.x == 0) s_W = Wgood;
        pos += (size_t)kGiantWaves * (64 * kGU);
      }
      if (ended) break;
    }
    __syncthreads();
#ifdef VBX_FOLD_STATS
    if (threadIdx.x == 0) { atomicMax(&st->dbg[5], dbg_rounds); atomicMax(&st->dbg[6], (uint32_t)(pos - i0)); }
#endif
    if (threadIdx.x == 0) {
      m.dist[gid] = s_d;
      m.weight[gid] = s_W;
      m.rgba[gid] = s_col;
    }
  }
}

// Long and giant runs in one launch (they are different voxels): the first `giant_blocks` workgroups take
// the giant runs — the sensor's own voxel keeps one of them busy for most of the kernel, so they start
// first — and the waves of all the others share the long ones.
// (8 waves per SIMD = 64 VGPRs, two workgroups per CU: measured 832 against 932 us with the default 81.)
__global__ void __launch_bounds__(64 * kGiantWaves, 8) k_fold_runs(const uint64_t* __restrict__ keys, size_t n, CastCfg c,
                                                                MapDev m, const float* __restrict__ in_sdf,
                                                                const float* __restrict__ in_uw,
                                                                const uint32_t* __restrict__ in_col,
                                                                const uint32_t* __restrict__ long_list, uint32_t long_cap,
                                                                const uint32_t* __restrict__ giant_list,
                                                                const uint8_t* __restrict__ ident, uint32_t giant_blocks,
                                                                DevState* st) {
  if (blockIdx.x < giant_blocks)
    fold_giant_runs(blockIdx.x, giant_blocks, keys, n, c, m, in_sdf, in_uw, in_col, giant_list, ident, st);
  else
    fold_long_runs((blockIdx.x - giant_blocks) * kGiantWaves + (threadIdx.x >> 6),
                   (gridDim.x - giant_blocks) * kGiantWaves, keys, n, c, m, in_sdf, in_uw, in_col, long_list, long_cap, st);
}

// ---------------------------------------------------------------------------
// kernels: Merged integrator bundling (tsdf_integrator.cc:340-407)
// ---------------------------------------------------------------------------
// One workgroup folds the per-workgroup boxes of k_prep_points into DevState::bbox_min / bbox_max.
__global__ void k_bbox_reduce(const int32_t* __restrict__ partial, uint32_t nblocks, DevState* st) {
  int lo[3] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF}, hi[3] = {-0x7FFFFFFF, -0x7FFFFFFF, -0x7FFFFFFF};
  for (uint32_t b = threadIdx.x; b < nblocks; b += blockDim.x)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = min(lo[a], partial[6 * b + a]);
      hi[a] = max(hi[a], partial[6 * b + 3 + a]);
    }
  __shared__ int s_lo[4][3], s_hi[4][3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      lo[a] = min(lo[a], __shfl_xor(lo[a], d));
      hi[a] = max(hi[a], __shfl_xor(hi[a], d));
    }
    if ((threadIdx.x & 63) == 0) {
      s_lo[threadIdx.x >> 6][a] = lo[a];
      s_hi[threadIdx.x >> 6][a] = hi[a];
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int a = threadIdx.x;
    st->bbox_min[a] = min(min(s_lo[0][a], s_lo[1][a]), min(s_lo[2][a], s_lo[3][a]));
    st->bbox_max[a] = max(max(s_hi[0][a], s_hi[1][a]), max(s_hi[2][a], s_hi[3][a]));
  }
}

// Bundle key of an endpoint voxel relative to the cloud's bounding box (KeyFrame): clearing bit on top, then
// z, y, x; invalid points (all ones) sort last.  merged_key_voxel / merged_key_packed undo it.
__device__ inline uint64_t merged_key(const KeyFrame& f, const l3& g, bool clearing) {
  uint64_t k = (uint64_t)(g.x - f.xmin) | ((uint64_t)(g.y - f.ymin) << f.bx) | ((uint64_t)(g.z - f.zmin) << (f.bx + f.by));
  if (clearing) k |= 1ull << (f.bx + f.by + f.bz);
  return k;
}
__device__ inline bool merged_key_clearing(const KeyFrame& f, uint64_t k) { return ((k >> (f.bx + f.by + f.bz)) & 1ull) != 0; }
__device__ inline l3 merged_key_voxel(const KeyFrame& f, uint64_t k) {
  return {(long long)(k & ((1ull << f.bx) - 1ull)) + f.xmin, (long long)((k >> f.bx) & ((1ull << f.by) - 1ull)) + f.ymin,
          (long long)((k >> (f.bx + f.by)) & ((1ull << f.bz) - 1ull)) + f.zmin};
}
// the voxel in the 3 x 21-bit packing the ray march compares against (anti-grazing, k_ray_emit)
__device__ inline uint64_t merged_key_packed(const KeyFrame& f, uint64_t k) {
  const l3 g = merged_key_voxel(f, k);
  return ((uint64_t)(g.z + (1ll << 20)) << 42) | ((uint64_t)(g.y + (1ll << 20)) << 21) | (uint64_t)(g.x + (1ll << 20));
}
__global__ void k_merged_keys(RayTab pt, uint32_t n, MapDev m, KeyFrame f, uint64_t* keys, uint32_t* vals) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const uint8_t fl = pt.flags[s];
  uint64_t key = ~0ull;
  if (fl & 1) key = merged_key(f, grid_index_from_point({pt.px[s], pt.py[s], pt.pz[s]}, m.voxel_size_inv), (fl & 2) != 0);
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

// Bundle fold (tsdf_integrator.cc:387-407): running weighted mean of point_C, blended colour, summed
// weight over the bundle's points in push_back (= visiting) order; clearing bundles take their first
// usable point only; merged_point_G = T_G_C * merged_point_C.
// bstart[rank] = position of the bundle's first point in the sorted (key, s) list
__global__ void k_merged_starts(const uint32_t* __restrict__ head, const uint32_t* __restrict__ rank, uint32_t n,
                                uint32_t* bstart) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && head[i]) bstart[rank[i]] = i;
}
// point data in sorted (key, s) order, so that a bundle's points are consecutive in memory
__global__ void k_merged_gather(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t n,
                                RayTab pt, const float* __restrict__ pcx, const float* __restrict__ pcy,
                                const float* __restrict__ pcz, float* gw, float* gx, float* gy, float* gz, uint32_t* gc) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n || keys[j] == ~0ull) return;
  const uint32_t s = vals[j];
  gw[j] = pt.w[s];
  gx[j] = pcx[s];
  gy[j] = pcy[s];
  gz[j] = pcz[s];
  gc[j] = pt.rgba[s];
}
// The fold with eight lanes per bundle: the three coordinates of the running mean and the four
// colour channels are seven independent chains that only share the running weight (Color::
// blendTwoColors works per channel, common.h:105-125), so each lane carries one of them, and the
// eight lanes fetch eight consecutive points at a time (one coalesced read instead of a chain of
// dependent gathers per point; the longest bundle of a frame — 100 to 300 points when the
// camera is close to a wall — used to cost ~0.85 us per point).
__global__ void k_merged_bundle8(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ bstart, uint32_t nb,
                                 uint32_t n, const float* __restrict__ gw, const float* __restrict__ gx,
                                 const float* __restrict__ gy, const float* __restrict__ gz,
                                 const uint32_t* __restrict__ gc, Pose T, KeyFrame kf, RayTab out, uint64_t* graze_keys,
                                 const uint32_t* __restrict__ perm) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t br = t >> 3;  // rank in ascending key order
  const int sub = (int)(t & 7u);
  const int g0 = (int)(threadIdx.x & 63u) & ~7;
  const bool live = br < nb;
  const uint32_t i0 = live ? bstart[br] : 0u;
  const uint64_t key = live ? keys[i0] : ~0ull;
  const bool clearing = live && merged_key_clearing(kf, key);
  float m = 0.0f;   // lanes 0-2: mean coordinate; lanes 3-6: colour channel r, g, b, a as a float
  float mw = 0.0f;
  bool more = live;
  for (uint32_t base = i0; __any(more); base += 8) {
    const uint32_t j = base + (uint32_t)sub;
    const bool mine = more && j < n && keys[j] == key;
    float pw = 0.f, x = 0.f, y = 0.f, z = 0.f;
    uint32_t col = 0;
    if (mine) {
      pw = gw[j]; x = gx[j]; y = gy[j]; z = gz[j]; col = gc[j];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const bool vq = __shfl((int)mine, g0 + q, 64) != 0;
      const float pwq = __shfl(pw, g0 + q, 64);
      const float xq = __shfl(x, g0 + q, 64), yq = __shfl(y, g0 + q, 64), zq = __shfl(z, g0 + q, 64);
      const uint32_t cq = (uint32_t)__shfl((int)col, g0 + q, 64);
      if (!more) continue;
      if (!vq) {  // the bundle ended inside this chunk
        more = false;
        continue;
      }
      if (pwq < 1e-6f) continue;
      const float tw = mw + pwq;
      if (sub < 3) {
        const float cmp = sub == 0 ? xq : (sub == 1 ? yq : zq);
        m = (m * mw + cmp * pwq) / tw;
      } else if (sub < 7) {
        const float w1 = mw / tw, w2 = pwq / tw;
        const float b = (float)(int)((cq >> (8 * (sub - 3))) & 0xFFu);
        m = (float)((int)roundf(m * w1 + b * w2) & 0xFF);
      }
      mw += pwq;
      if (clearing) more = false;  // clearing bundles take their first usable point only (:401-404)
    }
  }
  // gather the seven results in the group's first lane
  const float mx = __shfl(m, g0, 64), my = __shfl(m, g0 + 1, 64), mz = __shfl(m, g0 + 2, 64);
  const uint32_t cr = (uint32_t)(int)__shfl(m, g0 + 3, 64), cg = (uint32_t)(int)__shfl(m, g0 + 4, 64);
  const uint32_t cb = (uint32_t)(int)__shfl(m, g0 + 5, 64), ca = (uint32_t)(int)__shfl(m, g0 + 6, 64);
  if (!live || sub != 0) return;
  const uint32_t b = perm ? perm[br] : br;  // row = position in the visiting order of the bundles
  const f3 pg = pose_transform(T, f3{mx, my, mz});
  out.px[b] = pg.x;
  out.py[b] = pg.y;
  out.pz[b] = pg.z;
  out.rgba[b] = cr | (cg << 8) | (cb << 16) | (ca << 24);
  out.w[b] = mw;
  out.flags[b] = 1 | (clearing ? 2 : 0);
  const uint64_t packed = merged_key_packed(kf, key);
  out.bkey[b] = packed;
  if (!clearing && graze_keys) graze_keys[br] = packed;  // (z,y,x) order either way: stays sorted for the march's binary search
}

// Rows of the bundle table from key order (as folded) into the visiting order of the bundles: the fold runs
// while the host still reconstructs that order (vbx_host_tsdf.hpp), the permutation arrives afterwards.
__global__ void k_merged_permute_rows(RayTab in, const uint32_t* __restrict__ perm, uint32_t nb, RayTab out) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  const uint32_t r = perm[b];
  out.px[r] = in.px[b];
  out.py[r] = in.py[b];
  out.pz[r] = in.pz[b];
  out.rgba[r] = in.rgba[b];
  out.w[r] = in.w[b];
  out.flags[r] = in.flags[b];
  out.bkey[r] = in.bkey[b];
}

// The order in which bundleRays inserts the bundle keys into its unordered_map = ascending visiting
// position of each bundle's first point.  by_s[s] = bundle (ascending key rank) whose first point is
// visiting position s (else ~0); bpack[b] = clearing << 63 | b << 32 | LongIndexHash(voxel).
__global__ void k_merged_mark_first(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                    const uint32_t* __restrict__ head, const uint32_t* __restrict__ rank, uint32_t n,
                                    KeyFrame kf, uint32_t* by_s, uint64_t* bpack) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !head[i]) return;
  const uint32_t b = rank[i];
  const uint64_t k = keys[i];
  by_s[vals[i]] = b;
  const l3 g = merged_key_voxel(kf, k);
  bpack[b] = (merged_key_clearing(kf, k) ? (1ull << 63) : 0ull) | ((uint64_t)b << 32) |
             (uint64_t)long_index_hash(g);  // block_hash.h:54-64
}
__global__ void k_merged_first_flags(const uint32_t* __restrict__ by_s, uint32_t n, uint32_t* f) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > n) return;
  f[s] = (s < n && by_s[s] != 0xFFFFFFFFu) ? 1u : 0u;
}
__global__ void k_merged_insertion_order(const uint32_t* __restrict__ by_s, const uint32_t* __restrict__ pos,
                                         const uint64_t* __restrict__ bpack, uint32_t n, uint64_t* out) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const uint32_t b = by_s[s];
  if (b != 0xFFFFFFFFu) out[pos[s]] = bpack[b];
}

}  // namespace

