// vbx_kernels_esdf_strict.hpp — EsdfIntegrator in the REFERENCE'S OWN ORDER (cfg.reference_order = 1)
// Part of libvbx_hip.so's single translation unit (included by vbx_hip.hip, in order).
//
// The default ESDF path (vbx_kernels_esdf.hpp) computes order-free fixed points.  The reference's result is
// defined by its visiting order wherever min_diff_m > 0 or the sign-mismatch rule fires: updateFromTsdfBlocks
// walks the listed blocks voxel by voxel pushing into a FIFO raise queue and a BucketQueue (esdf_integrator.cc:
// 124-302), processRaiseSet pops the FIFO (:305-369), processOpenSet pops the lowest non-empty bucket, relaxes
// the 26 neighbours in LUT order and re-queues them (:371-496, bucket_queue.h:41-80), updateVoxelFromNeighbors
// pulls a new voxel's first estimate from the first LUT neighbour that beats it, with the LUT distance NOT scaled
// by the voxel size (:498-530).  This kernel replays exactly that, statement by statement, on the device arrays:
// ONE wave walks the sequence; what the reference does in a `for` over the 26 neighbours of one voxel (distinct
// voxels, independent of each other) is done by 26 lanes at once, what it does in a `for` over the voxels of a
// block is done 64 voxels at a time and committed in index order; every queue push keeps the reference's order
// (LUT order within a pop, voxel order within a block).  It is a sequential algorithm run at one wave's speed —
// the opt-in for callers who need the reference's bits, not the fast path.
//
// Queues: FIFOs of voxel ids (pool slot * vps^3 + linear index) in the chunked arena the parallel replay uses
// (vbx_esdf_replay_core.hpp: a FIFO index i of queue q lives in arena[chunk_tab[q][i / 1024] * 1024 + i % 1024], chunks
// are handed out by a bump counter and never reused inside an update); queue num_buckets is raise_, 0 .. num_buckets - 1
// are the buckets of open_.  With stop_before_open this kernel runs updateFromTsdfBlocks' voxel loop only and leaves
// raise_ and open_ to the replay; otherwise it also pops open_ one voxel at a time (the round-3
// form, kept for A/B checks: VBX_ESDF_REPLAY=0).

namespace {
constexpr uint32_t kSqChunk = 1024;
constexpr uint32_t kSqNone = 0xFFFFFFFFu;
constexpr int kStrictMaxBuckets = 255;

struct StrictArgs {
  MapDev m;
  EsdfDev e;
  EsdfCfgDev c;
  int full;         // Config::full_euclidean_distance
  int multi_queue;  // Config::multi_queue
  int num_buckets;  // Config::num_buckets (open_.setNumBuckets(num_buckets, max_distance_m), esdf_integrator.cc:21)
  int incremental;
  int batch_crust;  // !incremental && add_occupied_crust
  const uint32_t* list_slots;  // pool slots of the listed TSDF blocks in visiting order (kInvalidSlot: no such TSDF block)
  uint32_t n_list;
  uint32_t* arena;
  uint32_t* chunk_tab;     // [num_buckets + 1][n_chunks]
  uint32_t n_chunks;
  rp::Ctl* rctl;           // the replay's control block: FIFO heads / tails are handed over here
  int stop_before_open;
  unsigned long long* stats;  // [0] lower [1] raise [2] new [3] raised pops [4] open pops [5] relaxations [6] blocks [7] error
  unsigned long long max_pops;
};

// loads of anything this kernel may have written earlier bypass the vector L1 (served by the XCD's L2, where the
// wave's own stores have landed once drained)
__device__ inline float ld_f32(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline uint32_t ld_u32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void drain_stores() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }

struct StrictQueues {
  // wave-uniform state in LDS (every lane writes the same value; reads are broadcasts)
  volatile uint32_t head[kStrictMaxBuckets + 1], tail[kStrictMaxBuckets + 1];            // FIFO indices
  volatile uint32_t head_chunk[kStrictMaxBuckets + 1], tail_chunk[kStrictMaxBuckets + 1];  // arena chunk of head / tail - 1
  volatile uint32_t bump, err;
  volatile int last_bucket;           // BucketQueue::last_bucket_index_
  volatile unsigned long long n_open;  // BucketQueue::num_elements_
};

// wave-uniform push / pop (every lane calls with the same arguments)
__device__ inline void sq_push(StrictQueues& q, const StrictArgs& a, int qi, uint32_t gid) {
  if (q.err) return;
  const uint32_t idx = q.tail[qi];
  if ((idx % kSqChunk) == 0) {
    if (q.bump >= a.n_chunks) { q.err = 1; return; }  // arena exhausted: the entry is dropped, the host reports VBX_ERR_CAPACITY
    const uint32_t c = q.bump;
    q.bump = c + 1;
    if ((threadIdx.x & 63) == 0) a.chunk_tab[(size_t)qi * a.n_chunks + idx / kSqChunk] = c;
    q.tail_chunk[qi] = c;
    if (idx == q.head[qi]) q.head_chunk[qi] = c;
  }
  if ((threadIdx.x & 63) == 0) a.arena[(size_t)q.tail_chunk[qi] * kSqChunk + idx % kSqChunk] = gid;
  q.tail[qi] = idx + 1;
}
__device__ inline uint32_t sq_count(const StrictQueues& q, int qi) { return q.tail[qi] - q.head[qi]; }
__device__ inline uint32_t sq_pop(StrictQueues& q, const StrictArgs& a, int qi) {
  drain_stores();
  const uint32_t idx = q.head[qi];
  if ((idx % kSqChunk) == 0 && idx != 0) q.head_chunk[qi] = ld_u32(&a.chunk_tab[(size_t)qi * a.n_chunks + idx / kSqChunk]);
  const uint32_t gid = ld_u32(&a.arena[(size_t)q.head_chunk[qi] * kSqChunk + idx % kSqChunk]);
  q.head[qi] = idx + 1;
  return gid;
}
// BucketQueue::push (bucket_queue.h:41-56): the bucket of a value
__device__ inline int bucket_of(const StrictArgs& a, float value_f) {
  double value = (double)value_f;
  const double max_val = (double)a.c.max_distance;
  if (value > max_val) value = max_val;
  int b = (int)floor(fabs(value) / max_val * (double)(a.num_buckets - 1));
  if (b >= a.num_buckets) b = a.num_buckets - 1;
  if (b < 0) b = 0;  // (NaN / negative max_val cannot index a std::vector either)
  return b;
}
__device__ inline void open_push(StrictQueues& q, const StrictArgs& a, uint32_t gid, float value) {
  const int b = bucket_of(a, value);
  if (b < q.last_bucket) q.last_bucket = b;
  sq_push(q, a, b, gid);
  q.n_open = q.n_open + 1;
}
// BucketQueue::front + pop (:58-80)
__device__ inline uint32_t open_pop(StrictQueues& q, const StrictArgs& a) {
  int lb = q.last_bucket;
  while (lb < a.num_buckets && sq_count(q, lb) == 0) ++lb;
  q.last_bucket = lb;
  q.n_open = q.n_open - 1;
  return sq_pop(q, a, lb);
}

// getVoxelPtrByGlobalIndex(neighbour) for the voxel `off` away from (slot, lx, ly, lz): its id, or kSqNone when the
// ESDF layer has no block there (layer.h:222-239)
__device__ inline uint32_t neighbour_gid(const StrictArgs& a, uint32_t slot, int lx, int ly, int lz, const int off[3]) {
  const int vps = (int)a.m.vps;
  int nx = lx + off[0], ny = ly + off[1], nz = lz + off[2];
  int cx = 0, cy = 0, cz = 0;
  if (nx < 0) { nx += vps; cx = -1; } else if (nx >= vps) { nx -= vps; cx = 1; }
  if (ny < 0) { ny += vps; cy = -1; } else if (ny >= vps) { ny -= vps; cy = 1; }
  if (nz < 0) { nz += vps; cz = -1; } else if (nz >= vps) { nz -= vps; cz = 1; }
  uint32_t s2 = slot;
  if (cx | cy | cz) {
    s2 = map_find(a.m, pack_block_key(a.m.blk_idx[3 * slot] + cx, a.m.blk_idx[3 * slot + 1] + cy, a.m.blk_idx[3 * slot + 2] + cz));
    if (s2 == kInvalidSlot) return kSqNone;
  }
  if (!(ld_u32(&a.m.blk_flags[s2]) & kFlagEsdfAlloc)) return kSqNone;
  return s2 * a.m.nvox + (uint32_t)(nx + vps * (ny + vps * nz));
}

__device__ inline float lut_distance(int idx) {
  const float sq2 = (float)1.4142135623730951, sq3 = (float)1.7320508075688772;  // std::sqrt(2), std::sqrt(3) as float, neighbor_tools.cc:9-10
  return idx < 6 ? 1.0f : (idx < 18 ? sq2 : sq3);
}
// Eigen Vector3i::cast<float>().norm(): sqrt of the 3-element squaredNorm reduction c0 + (c1 + c2)
__device__ inline float parent_norm(int x, int y, int z) {
  const float fx = (float)x, fy = (float)y, fz = (float)z;
  return sqrtf(fx * fx + (fy * fy + fz * fz));
}

// the wave pushes the entries of the lanes in `mask`, in lane order (= LUT order / voxel order)
__device__ inline void push_lanes(StrictQueues& q, const StrictArgs& a, unsigned long long mask, int raise_q, bool to_raise,
                                  uint32_t gid, float value) {
  while (mask) {
    const int l = __ffsll((long long)mask) - 1;
    mask &= mask - 1;
    const uint32_t g = __shfl(gid, l);
    const float v = __shfl(value, l);
    if (to_raise) sq_push(q, a, raise_q, g); else open_push(q, a, g, v);
  }
}

__global__ void __launch_bounds__(64) k_esdf_strict(StrictArgs a) {
  __shared__ StrictQueues q;
  const int lane = threadIdx.x & 63;
  const int RQ = a.num_buckets;  // the raise queue's slot
  // the queues start with what addNewRobotPosition left in them (the host put the entries, the chunk table rows and the
  // tails in place; all zero otherwise)
  for (int i = lane; i <= a.num_buckets; i += 64) {
    const uint32_t t0 = a.rctl->tail[i];
    q.head[i] = 0; q.tail[i] = t0;
    q.head_chunk[i] = t0 ? a.chunk_tab[(size_t)i * a.n_chunks] : 0;
    q.tail_chunk[i] = t0 ? a.chunk_tab[(size_t)i * a.n_chunks + (t0 - 1) / kSqChunk] : 0;
  }
  __syncthreads();
  if (lane == 0) {
    unsigned long long n0 = 0;
    for (int i = 0; i < a.num_buckets; ++i) n0 += q.tail[i];
    q.bump = a.rctl->chunk_top; q.err = 0; q.last_bucket = 0; q.n_open = n0;
  }
  __syncthreads();
  const MapDev& m = a.m;
  const EsdfDev& e = a.e;
  const EsdfCfgDev& c = a.c;
  const int vps = (int)m.vps;
  unsigned long long n_lower = 0, n_raise = 0, n_new = 0, n_raised = 0, n_pops = 0, n_relax = 0, n_blocks = 0;

  // ---- updateFromTsdfBlocks, esdf_integrator.cc:136-287 -------------------------------------------------
  for (uint32_t bi = 0; bi < a.n_list && !q.err; ++bi) {
    const uint32_t slot = a.list_slots[bi];
    if (slot == kInvalidSlot) continue;                                  // :139-143 no such TSDF block
    if (!(ld_u32(&m.blk_flags[slot]) & kFlagPublished)) continue;
    // :145-147 allocateBlockPtrByIndex + set_updated(true) (kMap only)
    if (lane == 0) atomicOr(&m.blk_flags[slot], kFlagEsdfAlloc | (1u << kFlagEsdfUpdShift) | kFlagEsdfDirty);
    drain_stores();
    ++n_blocks;
    for (uint32_t base = 0; base < m.nvox; base += 64) {
      const uint32_t lin = base + lane;
      const uint32_t gid = slot * m.nvox + lin;
      const float td = m.dist[gid];
      const float tw = m.weight[gid];
      float ed = ld_f32(&e.dist[gid]);
      uint32_t es = ld_u32(&e.state[gid]);
      bool write = false, push_open = false, push_raise = false, needs_nb = false;
      if (tw < c.min_weight) {
        if (a.batch_crust) {  // :154-161
          ed = -c.default_distance;
          es = (es | kEsdfObserved | kEsdfHallucinated) & ~kEsdfFixed;
          write = true;
        }
      } else {
        write = true;
        const bool tsdf_fixed = fabsf(td) < c.min_distance;
        const float sgn_default = (float)signum(td) * c.default_distance;
        if (!(es & kEsdfObserved) || (es & kEsdfHallucinated)) {           // :173-199
          if (es & kEsdfHallucinated) push_raise = true;
          if (tsdf_fixed) {
            ed = td;
            es |= kEsdfFixed | kEsdfInQueue;
            push_open = true;
          } else {
            ed = sgn_default;
            es &= ~kEsdfFixed;
            needs_nb = a.incremental != 0;
          }
          es &= 0xFFu;  // parent.setZero() (also after updateVoxelFromNeighbors set it, :197)
          ++n_new;
        } else {
          const bool efixed = (es & kEsdfFixed) != 0;
          if (tsdf_fixed || efixed) {
            if (!tsdf_fixed) {                                             // :211-220
              ed = sgn_default;
              es = (es & 0xFFu & ~kEsdfFixed) | kEsdfInQueue;
              push_raise = true; push_open = true;
              ++n_raise;
            } else if ((ed > 0.0f && td + c.min_diff < ed) || (ed <= 0.0f && td - c.min_diff > ed)) {  // lower :221-237
              ed = td;
              es = (es & 0xFFu) | kEsdfFixed | kEsdfInQueue;
              push_open = true;
              ++n_lower;
            } else if ((ed > 0.0f && td - c.min_diff > ed) || (ed <= 0.0f && td + c.min_diff < ed)) {  // raise :238-256
              ed = td;
              es = (es & 0xFFu) | kEsdfFixed | kEsdfInQueue;
              push_raise = true; push_open = true;
              ++n_raise;
            }
          } else if (signum(td) != signum(ed)) {                           // :257-277
            if (td < ed) {
              ed = sgn_default;
              es = (es & 0xFFu) | kEsdfInQueue;
              push_open = true;
              ++n_lower;
            } else {
              ed = sgn_default;
              es &= 0xFFu;
              push_raise = true;
              ++n_raise;
            }
          }
        }
        es |= kEsdfObserved;          // :282-283
        es &= ~kEsdfHallucinated;
      }
      // commit in voxel order.  Only updateVoxelFromNeighbors (incremental, new voxel outside the fixed band) reads
      // OTHER voxels: everything in front of such a voxel must be written before it looks, nothing behind it may be.
      const unsigned long long nbm = __ballot(needs_nb);
      unsigned long long done_mask = 0;
      for (;;) {
        const unsigned long long pend_nb = nbm & ~done_mask;
        const int stop = pend_nb ? (__ffsll((long long)pend_nb) - 1) : 64;
        const unsigned long long seg = (stop == 64 ? ~0ull : ((1ull << stop) - 1ull)) & ~done_mask;
        if (write && ((seg >> lane) & 1ull)) {
          e.dist[gid] = ed;
          e.state[gid] = es;
        }
        // pushes of the segment, voxel by voxel: raise_ then open_ are separate queues, their relative order is free
        push_lanes(q, a, __ballot(push_raise) & seg, RQ, true, gid, ed);
        push_lanes(q, a, __ballot(push_open) & seg, RQ, false, gid, ed);
        done_mask |= seg;
        if (stop == 64) break;
        drain_stores();
        // ---- updateVoxelFromNeighbors(global_index) for lane `stop`'s voxel, :498-530 (26 lanes = 26 neighbours)
        const uint32_t vg = __shfl(gid, stop);
        const float vd = __shfl(ed, stop);      // sign * default
        uint32_t vs = __shfl(es, stop);
        if (__shfl((int)push_raise, stop)) sq_push(q, a, RQ, vg);  // :174-176, the voxel was hallucinated
        const uint32_t vlin = vg - slot * m.nvox;
        const int lx = (int)(vlin % vps), ly = (int)((vlin / vps) % vps), lz = (int)(vlin / (vps * vps));
        bool hit = false;
        float nd = 0.f;
        if (lane < 26) {
          const uint32_t ng = neighbour_gid(a, slot, lx, ly, lz, c_nb_off[lane]);
          if (ng != kSqNone) {
            const uint32_t ns = ld_u32(&e.state[ng]);
            nd = ld_f32(&e.dist[ng]);
            if ((ns & kEsdfObserved) && !(nd >= c.max_distance || nd <= -c.max_distance))
              hit = (signum(nd) == signum(vd)) && (fabsf(nd) < fabsf(vd));
          }
        }
        const unsigned long long hm = __ballot(hit);
        float new_d = vd;
        if (hm) {
          const int idx = __ffsll((long long)hm) - 1;
          const float d_n = __shfl(nd, idx);
          new_d = d_n + (float)signum(vd) * lut_distance(idx);  // NOT scaled by the voxel size (:508, :522)
          vs |= kEsdfInQueue;                                    // :187-190
          // voxel->parent = -(neighbor - global) is overwritten by parent.setZero() right after (:197)
        }
        if (lane == 0) {
          e.dist[vg] = new_d;
          e.state[vg] = vs;
        }
        if (hm) open_push(q, a, vg, new_d);
        done_mask |= 1ull << stop;
      }
    }
  }

  // ---- processRaiseSet, :305-369 -----------------------------------------------------------------------
  while (!a.stop_before_open && sq_count(q, RQ) != 0 && !q.err && n_raised + n_pops < a.max_pops) {
    const uint32_t g = sq_pop(q, a, RQ);
    const uint32_t slot = g / m.nvox, lin = g % m.nvox;
    const int lx = (int)(lin % vps), ly = (int)((lin / vps) % vps), lz = (int)(lin / (vps * vps));
    bool to_raise = false, to_open = false;
    uint32_t ng = kSqNone;
    float nd = 0.f;
    if (lane < 26) {
      ng = neighbour_gid(a, slot, lx, ly, lz, c_nb_off[lane]);
      if (ng != kSqNone) {
        uint32_t ns = ld_u32(&e.state[ng]);
        nd = ld_f32(&e.dist[ng]);
        if ((ns & kEsdfObserved) && !(ns & kEsdfFixed)) {
          int px, py, pz;
          unpack_parent(ns, &px, &py, &pz);
          const int dx = c_nb_off[lane][0], dy = c_nb_off[lane][1], dz = c_nb_off[lane][2];
          bool is_parent = (px == -dx && py == -dy && pz == -dz);
          if (a.full) {  // :340-348: the rounded normalised parent against the direction
            const float n2 = (float)px * (float)px + ((float)py * (float)py + (float)pz * (float)pz);
            float ux = (float)px, uy = (float)py, uz = (float)pz;
            if (n2 > 0.f) {
              const float nn = sqrtf(n2);
              ux = ux / nn; uy = uy / nn; uz = uz / nn;
            }
            is_parent = ((int)roundf(ux) == -dx && (int)roundf(uy) == -dy && (int)roundf(uz) == -dz);
          }
          if (is_parent) {
            nd = (float)signum(nd) * c.default_distance;
            e.dist[ng] = nd;
            e.state[ng] = ns & 0xFFu;  // parent.setZero()
            to_raise = true;
          } else if (!(ns & kEsdfInQueue)) {
            e.state[ng] = ns | kEsdfInQueue;
            to_open = true;
          }
        }
      }
    }
    if (lane < 26 && (to_raise || to_open)) atomicOr(&m.blk_flags[ng / m.nvox], kFlagEsdfDirty);
    // pushes in LUT order; a neighbour goes to exactly one of the two queues
    push_lanes(q, a, __ballot(to_raise), RQ, true, ng, nd);
    push_lanes(q, a, __ballot(to_open), RQ, false, ng, nd);
    ++n_raised;
  }

  if (a.stop_before_open) {
    // open_ goes to the parallel replay: FIFO indices and the arena's fill level into its control block
    drain_stores();
    for (int i = lane; i <= a.num_buckets; i += 64) {   // open_'s buckets and (index num_buckets) raise_
      a.rctl->head[i] = q.head[i];
      a.rctl->tail[i] = q.tail[i];
      a.rctl->reserved[i] = (q.tail[i] + kSqChunk - 1) / kSqChunk;
      a.rctl->k_cur[i] = 0;
    }
    if (lane == 0) a.rctl->chunk_top = q.bump;
  }
  // ---- processOpenSet, :371-496 ------------------------------------------------------------------------
  while (!a.stop_before_open && q.n_open != 0 && !q.err && n_raised + n_pops < a.max_pops) {
    const uint32_t g = open_pop(q, a);
    ++n_pops;
    uint32_t vs = ld_u32(&e.state[g]);
    const float vd = ld_f32(&e.dist[g]);
    if (lane == 0) e.state[g] = vs & ~kEsdfInQueue;  // :386
    vs &= ~kEsdfInQueue;
    if (!(vs & kEsdfObserved) || vd >= c.max_distance || vd <= -c.max_distance) continue;
    const uint32_t slot = g / m.nvox, lin = g % m.nvox;
    const int lx = (int)(lin % vps), ly = (int)((lin / vps) % vps), lz = (int)(lin / (vps * vps));
    int vpx, vpy, vpz;
    unpack_parent(vs, &vpx, &vpy, &vpz);
    bool push = false;
    uint32_t ng = kSqNone;
    float new_d = 0.f;
    if (lane < 26) {
      ng = neighbour_gid(a, slot, lx, ly, lz, c_nb_off[lane]);
      if (ng != kSqNone) {
        const uint32_t ns = ld_u32(&e.state[ng]);
        const float nd = ld_f32(&e.dist[ng]);
        if ((ns & kEsdfObserved) && !(ns & kEsdfFixed)) {
          const int dx = c_nb_off[lane][0], dy = c_nb_off[lane][1], dz = c_nb_off[lane][2];
          float distance = lut_distance(lane) * c.voxel_size;
          int npx = -dx, npy = -dy, npz = -dz;
          bool skip = false;
          if (a.full) {  // :419-428
            npx = vpx - dx; npy = vpy - dy; npz = vpz - dz;
            distance = c.voxel_size * (parent_norm(npx, npy, npz) - parent_norm(vpx, vpy, vpz));
            if ((double)distance < 0.0) skip = true;
          }
          bool upd = false;
          if (!skip) {
            if (vd > 0 && nd > 0) {                                        // :431-444
              if (vd + distance + c.min_diff < nd) { new_d = vd + distance; upd = true; }
            } else if (vd <= 0 && nd <= 0) {                               // :446-459
              if (vd - distance - c.min_diff > nd) { new_d = vd - distance; upd = true; }
            } else {                                                       // :461-491
              const float potential = vd - (float)signum(vd) * distance;
              if (fabsf(potential - nd) > distance) {
                if ((float)signum(potential) == nd) new_d = potential;     // :464 compares signum(int) with the float distance
                else new_d = (float)signum(nd) * distance;
                upd = true;
              }
            }
          }
          if (upd) {
            uint32_t s2 = (ns & 0xFFu) | pack_parent(npx, npy, npz);
            if (a.multi_queue || !(ns & kEsdfInQueue)) {
              push = true;
              s2 |= kEsdfInQueue;
            }
            e.dist[ng] = new_d;
            e.state[ng] = s2;
            atomicOr(&m.blk_flags[ng / m.nvox], kFlagEsdfDirty);
          }
          n_relax += upd ? 1 : 0;
        }
      }
    }
    push_lanes(q, a, __ballot(push), RQ, false, ng, new_d);
  }
  // per-lane counters: relaxations were counted by the lane that made them
  for (int d = 32; d > 0; d >>= 1) {  // (lower / raise / new were counted by the lane that classified the voxel)
    n_relax += __shfl_xor(n_relax, d);
    n_lower += __shfl_xor(n_lower, d);
    n_raise += __shfl_xor(n_raise, d);
    n_new += __shfl_xor(n_new, d);
  }
  if (lane == 0) {
    a.stats[0] = n_lower; a.stats[1] = n_raise; a.stats[2] = n_new; a.stats[3] = n_raised; a.stats[4] = n_pops;
    a.stats[5] = n_relax; a.stats[6] = n_blocks;
    a.stats[7] = q.err ? 1ull : ((!a.stop_before_open && (sq_count(q, RQ) != 0 || q.n_open != 0)) ? 2ull : 0ull);
  }
}

// The layer a reference-order (or full-Euclidean) update leaves is not a fixed point of the order-free relaxation: if the next
// update runs order-free, every ESDF block has to be relaxed as a whole, not only its shell (k_esdf_tile's shell_only pass)
__global__ void k_esdf_mark_unsettled(MapDev m, uint32_t used) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= used) return;
  const uint32_t f = m.blk_flags[s];
  if (!(f & kFlagFree) && (f & kFlagEsdfAlloc) && !(f & kFlagEsdfUnsettled)) atomicOr(&m.blk_flags[s], kFlagEsdfUnsettled);
}

// Update::kEsdf off on the listed TSDF blocks (updateFromTsdfLayer(clear_updated_flag = true), :113-121)
__global__ void k_esdf_strict_clear_tsdf_bit(MapDev m, const uint32_t* __restrict__ slots, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t s = slots[i];
  if (s != kInvalidSlot) atomicAnd(&m.blk_flags[s], ~4u);
}

}  // namespace
