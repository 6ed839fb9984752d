// vbx_kernels_esdf.hpp — EsdfIntegrator kernels: classification, robot spheres, LDS-tiled raise / lower / parent passes
// Part of libvbx_hip.so's single translation unit (included by vbx_hip.hip, in order).

namespace {
// ===========================================================================
// ESDF integrator (esdf_integrator.cc) — see DESIGN.md §4.5, HISTORY.md §4.4.
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
  uint32_t* active;   // per slot: 1 = process this sweep, 2 = process next sweep (full-Euclidean colour sweeps), 4 = touched,
                      // 8 / 16 / 32 = classification marks, 64 = written by the raise phase of this update, 128 = has been
                      // through the lower phase of this update (k_esdf_tile's first pass); bits 8..31 = the update-wide sweep number at which the block
                      // runs next (quasi-Euclidean sweeps: a tag instead of a rotate launch after every sweep)
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
    atomicOr(&m.blk_flags[slot], kFlagEsdfAlloc | (1u << kFlagEsdfUpdShift) | kFlagEsdfDirty);  // set_updated(true): kMap only
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
      want |= kFlagEsdfPendClassify | kFlagEsdfDirty;
    }
  } else {
    if (!(es & kEsdfObserved)) {
      e.dist[gid] = -default_distance;
      e.state[gid] = (es & 0xFFu) | kEsdfObserved | kEsdfHallucinated;
      want |= kFlagEsdfPendClassify | kFlagEsdfDirty;
    } else {
      want |= kFlagEsdfPendOpen;  // open_.push(global_index, distance) — in_queue stays clear (:81-85)
    }
  }
  const uint32_t cur = __hip_atomic_load(&m.blk_flags[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((cur & want) != want) atomicOr(&m.blk_flags[slot], want);
}

// The same two passes for reference_order: the voxel changes are applied here, but what the reference pushes into raise_ /
// open_ and inserts into updated_blocks_ goes back to the host per cube cell (gid, or kInvalidSlot outside the sphere;
// code: bit 0 the voxel changed (:54, :80), bit 1 raise_.push (:48), bit 2 open_.push (:84) with its bucket in bits 8-15):
// the host puts the entries in the iteration order of the reference's HierarchicalIndexMap (esdf_add_new_robot_position).
__global__ void k_sphere_apply_ordered(MapDev m, EsdfDev e, SphereDev sp, float default_distance, float max_distance, int num_buckets,
                                       int mode, uint32_t* __restrict__ out_gid, uint16_t* __restrict__ out_code) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)sp.n;
  if (t >= n * n * n) return;
  uint64_t key;
  uint32_t lin;
  uint32_t gid = kInvalidSlot;
  uint32_t code = 0;
  if (sphere_voxel(sp, m, t, &key, &lin)) {
    const uint32_t slot = map_find(m, key);
    if (slot != kInvalidSlot) {  // (pool exhausted otherwise: reported through DevState::error)
      gid = slot * m.nvox + lin;
      const uint32_t es = e.state[gid];
      bool changed = false;
      if (mode == 0) {
        if (!(es & kEsdfObserved) || (es & kEsdfHallucinated)) {
          if (es & kEsdfHallucinated) code |= 2u;
          e.dist[gid] = default_distance;
          changed = true;
        }
      } else {
        if (!(es & kEsdfObserved)) {
          e.dist[gid] = -default_distance;
          changed = true;
        } else if (!(es & kEsdfInQueue)) {
          // BucketQueue::push (bucket_queue.h:41-56)
          double value = (double)e.dist[gid];
          const double max_val = (double)max_distance;
          if (value > max_val) value = max_val;
          int b = (int)floor(fabs(value) / max_val * (double)(num_buckets - 1));
          if (b >= num_buckets) b = num_buckets - 1;
          if (b < 0) b = 0;
          code |= 4u | ((uint32_t)b << 8);
        }
      }
      if (changed) {
        e.state[gid] = (es & 0xFFu) | kEsdfObserved | kEsdfHallucinated;  // parent.setZero()
        code |= 1u;
        const uint32_t want = kFlagEsdfAlloc | kFlagEsdfDirty | kFlagEsdfUnsettled;
        const uint32_t cur = __hip_atomic_load(&m.blk_flags[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((cur & want) != want) atomicOr(&m.blk_flags[slot], want);
      }
    }
  }
  out_gid[t] = gid;
  out_code[t] = (uint16_t)code;
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
// block colour = parity of the block index per axis: blocks of one colour are never 26-adjacent
__device__ inline int esdf_block_colour(const MapDev& m, uint32_t slot) {
  return (m.blk_idx[3 * slot] & 1) | ((m.blk_idx[3 * slot + 1] & 1) << 1) | ((m.blk_idx[3 * slot + 2] & 1) << 2);
}
// the same after one colour's launch of the full-Euclidean lower phase: blocks of the other
// colours have not run yet and stay scheduled
__global__ void k_esdf_rotate_active_colour(MapDev m, EsdfDev e, uint32_t n_slots, int colour) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  const uint32_t a = e.active[s];
  uint32_t cur = (a & 2u) ? 1u : 0u;
  if ((m.blk_flags[s] & kFlagEsdfAlloc) && esdf_block_colour(m, s) != colour) cur |= (a & 1u);
  e.active[s] = cur | (a & 12u) | (cur ? 4u : 0u);
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
#ifdef VBX_ESDF_STATS  // measurement build (tools/esdf_tile_stats.py): what a lower-phase workgroup spends its time on
__device__ unsigned long long g_esdf_stats[16];
#endif
// FULL = Config::full_euclidean_distance: parents are accumulated vectors to the source voxel and a
// step costs voxel_size * (|parent - direction| - |parent|) (esdf_integrator.cc:419-428).  The
// result of that propagation depends on the path (the reference's on its queue order), so this
// variant is made race-free and deterministic instead of chaotic: the lower phase runs one block
// colour (parity of the block index, 8 colours) per launch, so no two adjacent blocks are relaxed
// concurrently, and inside the tile every iteration evaluates all queued voxels against the
// previous iteration's state before any of them is written (Jacobi).
// Sweeps of the quasi-Euclidean update are queued ahead of the host's convergence check (esdf_update_t): a launch
// carries the conditions under which it still makes sense — guard0 / guard1: the raise / lower phase must have gone
// idle before the sweep with that number (otherwise the phase has not converged in the sweeps queued for it: every
// workgroup leaves, the host finds out at its one read-back and continues sweep by sweep); force = addNewRobotPosition
// left raise marks the classification counters do not know about.
struct TileGuard {
  uint32_t guard0, guard1;
  int speculative, force;
};
template <int VPS, bool FULL>
__global__ void __launch_bounds__(kEsdfThreads) k_esdf_tile(MapDev m, EsdfDev e, EsdfCfgDev c, int mode,
                                                   uint32_t sweep_no, DevState* st, uint32_t g_sweep = 0, int first = 0,
                                                   TileGuard gd = TileGuard{0, 0, 0, 0}) {
  constexpr int T = VPS + 2;
  constexpr int NT = T * T * T;
  constexpr int NV = VPS * VPS * VPS;
  __shared__ float s_d[NT];
  __shared__ uint32_t s_s[NT];
  __shared__ uint8_t s_r[NT];     // raise marks (mode 0) / need flags (mode 1): never both
  __shared__ uint16_t s_q[NV];    // mode 1: dense work queue
  // mode 1, quasi-Euclidean: what a voxel offers its neighbours — its distance if it is observed and inside
  // +-max_distance, NaN otherwise (one LDS read and no validity test per neighbour in the relaxation)
  __shared__ float s_w[FULL ? 1 : NT];
  __shared__ int s_qn;
  __shared__ uint32_t s_nb[27];
  __shared__ int s_flag;
  const uint32_t slot = blockIdx.x;
  if (gd.speculative) {
    if (gd.guard0 && st->esdf_phase_changed[0] >= gd.guard0) return;
    if (gd.guard1 && st->esdf_phase_changed[1] >= gd.guard1) return;
    if (!gd.force && (st->esdf_blocks == 0 || (mode == 0 && !st->esdf_raise_any))) return;
  }
  if (!(m.blk_flags[slot] & kFlagEsdfAlloc)) return;
  const uint32_t act0 = e.active[slot];
  {
    const uint32_t a = act0;
    // g_sweep != 0: tagged scheduling — the block runs in the sweep it was tagged for, and in the first sweep of a
    // phase if anything touched it during this update; g_sweep == 0: the rotating flags of the colour sweeps
    if (g_sweep ? !((a >> 8) == g_sweep || (first && (a & 4u))) : !(a & 1u)) return;
  }
  if (FULL && mode == 1 && esdf_block_colour(m, slot) != (int)(sweep_no & 7u)) return;
  const int tid = threadIdx.x;
#ifdef VBX_ESDF_STATS
  const unsigned long long t_begin = wall_clock64();
  unsigned long long t_loaded = 0, n_iter = 0, n_eval = 0, t_compact = 0, t_relax = 0;
#endif
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
#ifdef VBX_ESDF_STATS
  t_loaded = wall_clock64();
#endif

  const float sq2 = (float)1.4142135623730951, sq3 = (float)1.7320508075688772;
  bool any_change = false;

  // processOpenSet for one voxel (pull form): the best value its neighbours offer; true if it
  // lowers the voxel.
  auto relax_eval = [&](int t, float* d_out, uint32_t* s_out) -> bool {
    const uint32_t s = s_s[t];
    if (!(s & kEsdfObserved) || (s & kEsdfFixed)) return false;
    float d = s_d[t];
    bool upd = false;
    uint32_t best_parent = 0;
    if (!FULL) {
      // The 26 neighbours in LUT order against the running value, as selects: the 52 LDS reads of a voxel issue in two
      // batches of 26 and nothing branches.  The loop is bound by its VALU work (a workgroup evaluates ~1000 voxels per
      // iteration on 4 SIMDs, tools/esdf_tile_stats.py), so the common case gets the short form: every usable
      // neighbour has the sign of the voxel (esdf_integrator.cc:437-458), computed on magnitudes — negation is exact, so
      // |v| + dist and the comparison give the bits of v - dist and its comparison.  A voxel with a usable neighbour
      // of the other sign takes the general form below from the start.
      const float step[3] = {1.0f * c.voxel_size, sq2 * c.voxel_size, sq3 * c.voxel_size};
      const bool dpos = d > 0.0f;
      const float sgn = dpos ? 1.0f : -1.0f;
      float D = d * sgn;
      float vmin = __builtin_inff();  // the smallest sign-adjusted offer: <= 0 (< 0 for d <= 0) is a usable neighbour of the other sign
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float w[13];
#pragma unroll
        for (int j = 0; j < 13; ++j) {
          const int i = 13 * h + j;
          w[j] = s_w[t + kNbOff[i][0] + T * (kNbOff[i][1] + T * kNbOff[i][2])];
        }
#pragma unroll
        for (int j = 0; j < 13; ++j) {
          const int i = 13 * h + j;
          const float V = w[j] * sgn;  // NaN stays NaN: no offer
          vmin = __builtin_fminf(vmin, V);
          const float C = V + step[i < 6 ? 0 : (i < 18 ? 1 : 2)];
          // min_diff_m does not gate the wavefront (the host passes 0: DESIGN.md 4.4), so the test is the bare comparison;
          // an offer of the other sign is found through vmin and redone in the general form
          const bool imp = C < D;
          D = imp ? C : D;
          best_parent = imp ? pack_parent(kNbOff[i][0], kNbOff[i][1], kNbOff[i][2]) : best_parent;  // -direction: toward the pusher
        }
      }
      const bool mismatch = dpos ? (vmin <= 0.0f) : (vmin < 0.0f);
      if (!mismatch) {
        *d_out = D * sgn;
        *s_out = (s & 0xFFu) | best_parent;
        return best_parent != 0;
      }
      best_parent = 0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        uint32_t sv[13];
        float dv[13];
#pragma unroll
        for (int j = 0; j < 13; ++j) {
          const int i = 13 * h + j;
          const int tv = t + kNbOff[i][0] + T * (kNbOff[i][1] + T * kNbOff[i][2]);
          sv[j] = s_s[tv];
          dv[j] = s_d[tv];
        }
#pragma unroll
        for (int j = 0; j < 13; ++j) {
          const int i = 13 * h + j;
          const float dist = step[i < 6 ? 0 : (i < 18 ? 1 : 2)];
          const float v = dv[j];
          const bool ok = (sv[j] & kEsdfObserved) && !(v >= c.max_distance || v <= -c.max_distance);
          const bool vpos = v > 0.0f, dp = d > 0.0f;
          // same sign (esdf_integrator.cc:437-458)
          const float cs = dp ? v + dist : v - dist;
          const bool imp_s = dp ? (cs + c.min_diff < d) : (cs - c.min_diff > d);
          // sign mismatch (:459-488) in its order-free form: see the FULL branch below for the rule
          const float sgv = vpos ? 1.0f : (v < 0.0f ? -1.0f : 0.0f);
          const float potential = v - sgv * dist;
          const float sgp = potential > 0.0f ? 1.0f : (potential < 0.0f ? -1.0f : 0.0f);
          const float sgd = dp ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
          const float cm = (sgp == d) ? potential : sgd * dist;
          const bool imp_m = fabsf(cm) < fabsf(d);
          const bool same = (vpos == dp);
          const bool imp = ok && (same ? imp_s : imp_m);
          d = imp ? (same ? cs : cm) : d;
          best_parent = imp ? pack_parent(kNbOff[i][0], kNbOff[i][1], kNbOff[i][2]) : best_parent;
          upd |= imp;
        }
      }
      *d_out = d;
      *s_out = (s & 0xFFu) | best_parent;
      return upd;
    }
#pragma unroll
    for (int i = 0; i < 26; ++i) {
      const int tv = t + kNbOff[i][0] + T * (kNbOff[i][1] + T * kNbOff[i][2]);
      const uint32_t sv = s_s[tv];
      if (!(sv & kEsdfObserved)) continue;
      const float dv = s_d[tv];
      if (dv >= c.max_distance || dv <= -c.max_distance) continue;
      float dist = (i < 6 ? 1.0f : (i < 18 ? sq2 : sq3)) * c.voxel_size;
      uint32_t parent = pack_parent(kNbOff[i][0], kNbOff[i][1], kNbOff[i][2]);  // -direction: toward the pusher
      {
        // new_parent = voxel->parent - direction; the step costs the growth of the parent vector
        int px, py, pz;
        unpack_parent(sv, &px, &py, &pz);
        const int nx = px + kNbOff[i][0], ny = py + kNbOff[i][1], nz = pz + kNbOff[i][2];
        const float nn = sqrtf((float)nx * (float)nx + ((float)ny * (float)ny + (float)nz * (float)nz));
        const float pn = sqrtf((float)px * (float)px + ((float)py * (float)py + (float)pz * (float)pz));
        dist = c.voxel_size * (nn - pn);
        if (dist < 0.0f) continue;
        parent = pack_parent(nx, ny, nz);
      }
      if (dv > 0 && d > 0) {
        if (dv + dist + c.min_diff < d) { d = dv + dist; best_parent = parent; upd = true; }
      } else if (dv <= 0 && d <= 0) {
        if (dv - dist - c.min_diff > d) { d = dv - dist; best_parent = parent; upd = true; }
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
        if (fabsf(cand) < fabsf(d)) { d = cand; best_parent = parent; upd = true; }
      }
    }
    *d_out = d;
    *s_out = (s & 0xFFu) | best_parent;
    return upd;
  };
  auto relax = [&](int t) -> bool {
    float d;
    uint32_t s;
    const bool upd = relax_eval(t, &d, &s);
    if (upd) {
      s_d[t] = d;
      s_s[t] = s;
      if (!FULL) s_w[t] = (fabsf(d) < c.max_distance) ? d : __builtin_nanf("");
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
    for (int t = tid; t < NT; t += kEsdfThreads) {
      s_need[t] = 0;
      if (!FULL) s_w[t] = ((s_s[t] & kEsdfObserved) && fabsf(s_d[t]) < c.max_distance) ? s_d[t] : __builtin_nanf("");
    }
    __syncthreads();
    // First pass: every voxel of a block whose interior this update has written (classification, robot spheres, the
    // raise phase) and that has not been through this phase yet.  Any other block stands at a local fixed point — of
    // its last run in this phase, or of the previous update, which ended at a global one — and only its halo can
    // have changed: the voxels next to the halo.
    const bool shell_only = !FULL && ((act0 & 128u) || !(act0 & (8u | 16u | 64u)));
    for (int v = tid; v < NV; v += kEsdfThreads) {
      const int lx = v % VPS, ly = (v / VPS) % VPS, lz = v / (VPS * VPS);
      const bool shell = lx == 0 || lx == VPS - 1 || ly == 0 || ly == VPS - 1 || lz == 0 || lz == VPS - 1;
      if (shell || !shell_only) s_need[(lx + 1) + T * ((ly + 1) + T * (lz + 1))] = 1;
    }
    bool converged = false;
    for (int iter = 0; iter < 64 * VPS; ++iter) {
#ifdef VBX_ESDF_STATS
      const unsigned long long t_it0 = wall_clock64();
#endif
      if (tid == 0) s_qn = 0;
      __syncthreads();
      {
        // a thread's voxels: flags and states read together, ONE queue reservation per thread
        constexpr int PER = (NV + kEsdfThreads - 1) / kEsdfThreads;
        int tt[PER];
        uint8_t nd[PER];
        uint32_t sv[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
          const int v = min(tid + k * kEsdfThreads, NV - 1);
          const int lx = v % VPS, ly = (v / VPS) % VPS, lz = v / (VPS * VPS);
          tt[k] = (lx + 1) + T * ((ly + 1) + T * (lz + 1));
          nd[k] = s_need[tt[k]];
          sv[k] = s_s[tt[k]];
        }
        // one reservation per WAVE: ballots give every entry its rank (a per-lane atomicAdd on the one counter
        // serialises 1024 LDS atomics per pass)
        const int lane = tid & 63;
        const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        unsigned long long mk[PER];
        int total = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
          if (tid + k * kEsdfThreads >= NV) nd[k] = 0;
          if (nd[k]) s_need[tt[k]] = 0;
          nd[k] = (nd[k] && (sv[k] & kEsdfObserved) && !(sv[k] & kEsdfFixed)) ? 1 : 0;
          mk[k] = __ballot(nd[k]);
          total += (int)__popcll(mk[k]);
        }
        if (total) {  // wave-uniform
          int at = 0;
          if (lane == 0) at = atomicAdd(&s_qn, total);
          at = __shfl(at, 0);
#pragma unroll
          for (int k = 0; k < PER; ++k) {
            if (nd[k]) s_q[at + (int)__popcll(mk[k] & lt)] = (uint16_t)tt[k];
            at += (int)__popcll(mk[k]);
          }
        }
      }
      __syncthreads();
      const int qn = s_qn;
      if (qn == 0) {
        converged = true;
        break;
      }
#ifdef VBX_ESDF_STATS
      ++n_iter;
      n_eval += (unsigned long long)qn;
      const unsigned long long t_it1 = wall_clock64();
      t_compact += t_it1 - t_it0;
#endif
      if (FULL) {
        // Jacobi: NV / kEsdfThreads queue entries per thread at most, evaluated against the state
        // of the previous iteration, written after the barrier
        constexpr int PER = (NV + kEsdfThreads - 1) / kEsdfThreads;
        float nd[PER];
        uint32_t ns[PER];
        bool up[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
          const int q = tid + k * kEsdfThreads;
          up[k] = q < qn && relax_eval(s_q[q], &nd[k], &ns[k]);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k) {
          if (!up[k]) continue;
          const int t = s_q[tid + k * kEsdfThreads];
          s_d[t] = nd[k];
          s_s[t] = ns[k];
          any_change = true;
#pragma unroll
          for (int i = 0; i < 26; ++i) s_need[t + kNbOff[i][0] + T * (kNbOff[i][1] + T * kNbOff[i][2])] = 1;
        }
      } else {
        for (int q = tid; q < qn; q += kEsdfThreads) {
          const int t = s_q[q];
          if (relax(t)) {
            any_change = true;
#pragma unroll
            for (int i = 0; i < 26; ++i) s_need[t + kNbOff[i][0] + T * (kNbOff[i][1] + T * kNbOff[i][2])] = 1;
          }
        }
      }
      __syncthreads();
#ifdef VBX_ESDF_STATS
      t_relax += wall_clock64() - t_it1;
#endif
    }
    if (!FULL && converged && !(act0 & 128u) && tid == 0) {
      atomicOr(&e.active[slot], 128u);
      if (!shell_only && (m.blk_flags[slot] & kFlagEsdfUnsettled)) atomicAnd(&m.blk_flags[slot], ~kFlagEsdfUnsettled);
    }
  } else {
  constexpr int PER = (NV + kEsdfThreads - 1) / kEsdfThreads;
  int tt[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int v = min(tid + k * kEsdfThreads, NV - 1);
    const int lx = v % VPS, ly = (v / VPS) % VPS, lz = v / (VPS * VPS);
    tt[k] = (lx + 1) + T * ((ly + 1) + T * (lz + 1));
  }
  if (mode == 0 && !FULL) {
    // Raise closure by pointer jumping.  A voxel the raise can reach (observed, not fixed, with a parent, not raised yet)
    // points at its parent voxel, every other voxel of the tile at itself; per round a voxel takes over the raise mark
    // of the voxel it points at and then points where that one pointed: a parent chain of length n inside the tile is
    // closed in log2(n) rounds instead of n.  (In place: a pointer only ever moves along the voxel's own chain, a mark
    // only ever comes from a voxel on it, so the order of the lanes does not matter; 13 rounds cover any chain or
    // cycle of the 18^3 tile.)
    uint16_t* s_nx = reinterpret_cast<uint16_t*>(s_w);
    for (int t = tid; t < NT; t += kEsdfThreads) s_nx[t] = (uint16_t)t;
    uint32_t sv[PER];
    float dv[PER];
    bool cand[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      sv[k] = s_s[tt[k]];
      dv[k] = s_d[tt[k]];
      cand[k] = s_r[tt[k]] == 0 && tid + k * kEsdfThreads < NV;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      int px, py, pz;
      unpack_parent(sv[k], &px, &py, &pz);
      cand[k] = cand[k] && (sv[k] & kEsdfObserved) && !(sv[k] & kEsdfFixed) && (px | py | pz) != 0;
      // quasi-Euclidean parents are unit LUT offsets, so the parent voxel is inside the halo
      if (cand[k]) s_nx[tt[k]] = (uint16_t)(tt[k] + px + T * (py + T * pz));
    }
    __syncthreads();
    for (int iter = 0; iter < 13; ++iter) {
      int nx[PER];
#pragma unroll
      for (int k = 0; k < PER; ++k) nx[k] = s_nx[tt[k]];
      uint8_t rp[PER];
      int nn[PER];
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        rp[k] = s_r[nx[k]];
        nn[k] = s_nx[nx[k]];
      }
      bool moved = false;
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        if (!cand[k]) continue;
        if (rp[k]) {
          s_d[tt[k]] = (float)signum(dv[k]) * c.default_distance;
          s_s[tt[k]] = sv[k] & 0xFFu;
          s_r[tt[k]] = 1;
          cand[k] = false;
          moved = true;
          any_change = true;  // only a raise changes the block: a moved pointer is scratch
        } else if (nn[k] != nx[k]) {
          s_nx[tt[k]] = (uint16_t)nn[k];
          moved = true;
        }
      }
      if (!__syncthreads_or(moved ? 1 : 0)) break;
    }
  } else if (mode == 0) {
    // raise closure: the reads of a thread's voxels issue together, then the reads of their parent voxels
    for (int iter = 0; iter < 4 * VPS; ++iter) {
      uint32_t sv[PER];
      float dv[PER];
      int tp[PER];
      bool cand[PER];
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        sv[k] = s_s[tt[k]];
        dv[k] = s_d[tt[k]];
        cand[k] = s_r[tt[k]] == 0 && tid + k * kEsdfThreads < NV;
      }
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        int px, py, pz;
        unpack_parent(sv[k], &px, &py, &pz);
        cand[k] = cand[k] && (sv[k] & kEsdfObserved) && !(sv[k] & kEsdfFixed) && (px | py | pz) != 0;
        if (FULL) {
          // esdf_integrator.cc:340-348: the parent *direction*, parent.normalized() rounded per
          // component (std::round), has to point at the raised voxel
          const f3 dir = f3_normalized(f3{(float)px, (float)py, (float)pz});
          px = (int)roundf(dir.x);
          py = (int)roundf(dir.y);
          pz = (int)roundf(dir.z);
          cand[k] = cand[k] && (px | py | pz) != 0;
        }
        // quasi-Euclidean parents are unit LUT offsets, so the parent voxel is inside the halo
        tp[k] = cand[k] ? tt[k] + px + T * (py + T * pz) : tt[k];
      }
      uint8_t rp[PER];
#pragma unroll
      for (int k = 0; k < PER; ++k) rp[k] = s_r[tp[k]];
      bool changed = false;
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        if (cand[k] && rp[k]) {
          s_d[tt[k]] = (float)signum(dv[k]) * c.default_distance;
          s_s[tt[k]] = sv[k] & 0xFFu;
          s_r[tt[k]] = 1;
          changed = true;
        }
      }
      any_change |= changed;
      if (!__syncthreads_or(changed ? 1 : 0)) break;
    }
  } else {
    // mode 2: canonical parent — the first LUT neighbour that explains the converged distance exactly; per voxel the
    // 52 reads in two batches, the first hit picked by selects
    const float step[3] = {1.0f * c.voxel_size, sq2 * c.voxel_size, sq3 * c.voxel_size};
    if (!FULL) {  // the neighbours' offers as in the lower phase
      for (int t = tid; t < NT; t += kEsdfThreads)
        s_w[t] = ((s_s[t] & kEsdfObserved) && fabsf(s_d[t]) < c.max_distance) ? s_d[t] : __builtin_nanf("");
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      if (tid + k * kEsdfThreads >= NV) continue;
      const int t = tt[k];
      const uint32_t s = s_s[t];
      if (!(s & kEsdfObserved) || (s & kEsdfFixed) || (s >> 8) == 0) continue;
      const float d = s_d[t];
      bool found = false;
      uint32_t ns = s;
      if (!FULL) {
        // short form (no usable neighbour of the other sign, see the lower phase): one read, a product, a sum, a
        // comparison and a select per neighbour; walked from the last LUT entry to the first, so the first hit stays
        const bool dpos = d > 0.0f;
        const float sgn = dpos ? 1.0f : -1.0f;
        const float D = d * sgn;
        float vmin = __builtin_inff();
        uint32_t sel = 0;
#pragma unroll
        for (int h = 1; h >= 0; --h) {
          float w[13];
#pragma unroll
          for (int j = 0; j < 13; ++j) {
            const int i = 13 * h + j;
            w[j] = s_w[t + kNbOff[i][0] + T * (kNbOff[i][1] + T * kNbOff[i][2])];
          }
#pragma unroll
          for (int j = 12; j >= 0; --j) {
            const int i = 13 * h + j;
            const float V = w[j] * sgn;
            vmin = __builtin_fminf(vmin, V);
            sel = (V + step[i < 6 ? 0 : (i < 18 ? 1 : 2)] == D) ? pack_parent(kNbOff[i][0], kNbOff[i][1], kNbOff[i][2]) : sel;
          }
        }
        if (!(dpos ? (vmin <= 0.0f) : (vmin < 0.0f))) {
          if (sel != 0 && ((s & 0xFFu) | sel) != s) {
            s_s[t] = (s & 0xFFu) | sel;
            any_change = true;
          }
          continue;
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        uint32_t sv[13];
        float dv[13];
#pragma unroll
        for (int j = 0; j < 13; ++j) {
          const int i = 13 * h + j;
          const int tv = t + kNbOff[i][0] + T * (kNbOff[i][1] + T * kNbOff[i][2]);
          sv[j] = s_s[tv];
          dv[j] = s_d[tv];
        }
#pragma unroll
        for (int j = 0; j < 13; ++j) {
          const int i = 13 * h + j;
          const float dist = step[i < 6 ? 0 : (i < 18 ? 1 : 2)];
          const float v = dv[j];
          const bool ok = (sv[j] & kEsdfObserved) && !(v >= c.max_distance || v <= -c.max_distance);
          const bool vpos = v > 0.0f, dpos = d > 0.0f;
          const float cs = dpos ? v + dist : v - dist;
          // the sign-mismatch rule: d == potential or sign(d) * dist
          const float sgv = vpos ? 1.0f : (v < 0.0f ? -1.0f : 0.0f);
          const float potential = v - sgv * dist;
          const float sgp = potential > 0.0f ? 1.0f : (potential < 0.0f ? -1.0f : 0.0f);
          const float sgd = dpos ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
          const float cm = (sgp == d) ? potential : sgd * dist;
          const bool hit = ok && (((vpos == dpos) ? cs : cm) == d);
          if (hit && !found) ns = (s & 0xFFu) | pack_parent(kNbOff[i][0], kNbOff[i][1], kNbOff[i][2]);
          found |= hit;
        }
      }
      if (ns != s) {
        s_s[t] = ns;
        any_change = true;
      }
    }
  }
  }
  if (any_change) s_flag = 1;
  __syncthreads();
#ifdef VBX_ESDF_STATS
  if (tid == 0 && mode == 1) {
    const unsigned long long t_end = wall_clock64();
    atomicAdd(&g_esdf_stats[0], 1ull);
    atomicAdd(&g_esdf_stats[1], n_iter);
    atomicMax(&g_esdf_stats[2], n_iter);
    atomicAdd(&g_esdf_stats[3], n_eval);
    atomicAdd(&g_esdf_stats[4], t_loaded - t_begin);
    atomicAdd(&g_esdf_stats[5], t_end - t_loaded);
    atomicMax(&g_esdf_stats[6], t_end - t_loaded);
    atomicMax(&g_esdf_stats[7], t_loaded - t_begin);
    if (s_flag) atomicAdd(&g_esdf_stats[8], 1ull);
    atomicMax(&g_esdf_stats[9], n_eval);
    atomicAdd(&g_esdf_stats[10], t_compact);
    atomicAdd(&g_esdf_stats[11], t_relax);
  }
#endif
  if (!s_flag) return;
  for (int v = tid; v < NV; v += kEsdfThreads) {
    const int lx = v % VPS, ly = (v / VPS) % VPS, lz = v / (VPS * VPS);
    const int t = (lx + 1) + T * ((ly + 1) + T * (lz + 1));
    const uint32_t g = slot * NV + v;
    e.dist[g] = s_d[t];
    e.state[g] = s_s[t];
    if (mode == 0) e.raised[g] = s_r[t];
  }
  if (tid == 0 && mode == 0 && !(act0 & 64u)) atomicOr(&e.active[slot], 64u);  // the raise phase wrote this block's interior
  if (tid == 0) atomicOr(&m.blk_flags[slot], kFlagEsdfDirty);  // the wavefront changed this block: the host mirror must take it
  if (mode != 2) {
    if (tid < 27 && s_nb[tid] != kInvalidSlot) {
      if (g_sweep) {  // tag the block and its neighbours for the next sweep (max of the tags: a later mark must not be lost)
        uint32_t* w = &e.active[s_nb[tid]];
        uint32_t old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (;;) {
          const uint32_t tag = max(old >> 8, g_sweep + 1u);
          const uint32_t want = (tag << 8) | (old & 0xFFu) | 4u;
          if (want == old) break;
          const uint32_t prev = atomicCAS(w, old, want);
          if (prev == old) break;
          old = prev;
        }
      } else {
        atomicOr(&e.active[s_nb[tid]], 2u | 4u);
      }
    }
    if (tid == 0) {
      atomicMax(&st->changed, sweep_no);  // the last sweep (1-based, per phase) in which a block changed
      if (g_sweep) atomicMax(&st->esdf_phase_changed[mode], g_sweep);
      atomicAdd(&st->esdf_relax_blocks, 1u);
    }
  }
}

}  // namespace

