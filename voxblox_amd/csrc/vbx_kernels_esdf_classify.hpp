// vbx_kernels_esdf_classify.hpp — updateFromTsdfBlocks' voxel loop (esdf_integrator.cc:136-287) in the reference's order, in parallel
// Part of libvbx_hip.so's single translation unit (included by vbx_hip.hip, in order).
//
// The reference walks the listed blocks one after the other, voxel by voxel, and pushes into raise_ / open_ as it goes.  What
// a voxel becomes depends on its own TSDF / ESDF voxel only — except for a NEW voxel outside the fixed band of an incremental
// update, which calls updateVoxelFromNeighbors (:498-530): it looks at its 26 neighbours AS THEY ARE AT THAT MOMENT (a
// neighbour that comes earlier in the walk has been classified already, a later one has not, a block listed later does not
// exist in the ESDF layer yet unless it did before the update) and takes its first estimate from the first LUT neighbour of
// its own sign that is closer — with the LUT distance NOT scaled by the voxel size (:508, Q9).  Here:
//   k_cls_base     thread per voxel of the listed blocks: the rule tree without the neighbour look, into shadow arrays
//                  (the layer itself stays as it was so that "not yet classified" can still be read)
//   k_cls_nb_block the voxels that look at neighbours: a neighbour in front of the voxel in the walk is read from the shadow, one
//                  behind it from the layer.  A voxel only depends on voxels in front of it (a cycle-free system); inside a
//                  block the sweep follows that order (hyperplanes), across block faces the launch is repeated until nothing
//                  moves.
//   k_cls_commit   shadow -> layer, block flags, queue of every push (raise_ / bucket of open_) and the totals per queue
//   k_cls_reserve  one thread: arena chunks for the totals, FIFO heads / tails into the replay's control block
//   k_cls_push     chained scans over the walk order: the pushes land in their FIFOs in the order the reference made them
// The one-wave form (k_esdf_strict) stays as the fallback for a list that names a block twice.

namespace {

constexpr uint32_t kClsCounters = 1024, kClsBase = 600, kClsScanTot = 1000;   // words of ClsArgs::counters; [kClsBase + q]: entries queue q held before the walk
constexpr uint32_t kClsWrite = 1, kClsOpen = 2, kClsRaise = 4, kClsNb = 8, kClsHit = 16;

struct ClsArgs {
  MapDev m;
  EsdfDev e;
  EsdfCfgDev c;
  int incremental, batch_crust, num_buckets;
  const uint32_t* list_slots;   // pool slot per list position (kInvalidSlot: no such TSDF block)
  uint32_t n_list;
  uint32_t* slot_pos;           // [pool slots] position in the list (kInvalidSlot: not listed)
  uint32_t* nb27;               // [n_list][27] pool slot of the neighbouring block (any layer), kInvalidSlot if none
  float* sh_d;                  // [n_list][nvox] shadow distance / state / flags / queue of the open_ push
  uint32_t* sh_s;
  uint8_t* sh_f;
  uint8_t* sh_q;
  uint32_t* counters;           // [0] duplicate list entries, [1] voxels whose neighbour look moved this round, [2] blocks walked, [8 ..] pushes per queue, [kClsBase ..] FIFO index of the walk's first push per queue
};

// a listed block that the walk really visits (:139-143)
__device__ inline bool cls_valid(const ClsArgs& a, uint32_t slot) {
  return slot != kInvalidSlot && (a.m.blk_flags[slot] & kFlagPublished);
}

__global__ void k_cls_positions(ClsArgs a) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.n_list) return;
  const uint32_t slot = a.list_slots[p];
  if (slot == kInvalidSlot) return;
  if (atomicMin(&a.slot_pos[slot], p) != kInvalidSlot) atomicAdd(&a.counters[0], 1u);   // listed twice: the one-wave form handles that
}
__global__ void k_cls_neighbours(ClsArgs a) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_list * 27u) return;
  const uint32_t p = i / 27u, k = i % 27u;
  const uint32_t slot = a.list_slots[p];
  uint32_t out = kInvalidSlot;
  if (slot != kInvalidSlot) {
    const int dx = (int)(k % 3) - 1, dy = (int)(k / 3 % 3) - 1, dz = (int)(k / 9) - 1;
    out = (dx | dy | dz) ? map_find(a.m, pack_block_key(a.m.blk_idx[3 * slot] + dx, a.m.blk_idx[3 * slot + 1] + dy, a.m.blk_idx[3 * slot + 2] + dz)) : slot;
  }
  a.nb27[i] = out;
}

// the rule tree of :149-284 for one voxel, without updateVoxelFromNeighbors
__global__ void k_cls_base(ClsArgs a) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)a.n_list * a.m.nvox) return;
  const uint32_t p = (uint32_t)(i / a.m.nvox), lin = (uint32_t)(i % a.m.nvox);
  const uint32_t slot = a.list_slots[p];
  uint8_t f = 0;
  float ed = 0.f;
  uint32_t es = 0;
  if (cls_valid(a, slot)) {
    if (lin == 0) atomicAdd(&a.counters[2], 1u);
    const EsdfCfgDev& c = a.c;
    const uint32_t gid = slot * a.m.nvox + lin;
    const float td = a.m.dist[gid], tw = a.m.weight[gid];
    ed = a.e.dist[gid];
    es = a.e.state[gid];
    if (tw < c.min_weight) {
      if (a.batch_crust) {  // :154-161
        ed = -c.default_distance;
        es = (es | kEsdfObserved | kEsdfHallucinated) & ~kEsdfFixed;
        f = kClsWrite;
      }
    } else {
      f = kClsWrite;
      const bool tsdf_fixed = fabsf(td) < c.min_distance;
      const float sgn_default = (float)signum(td) * c.default_distance;
      if (!(es & kEsdfObserved) || (es & kEsdfHallucinated)) {           // :173-199
        if (es & kEsdfHallucinated) f |= kClsRaise;
        if (tsdf_fixed) {
          ed = td;
          es |= kEsdfFixed | kEsdfInQueue;
          f |= kClsOpen;
        } else {
          ed = sgn_default;
          es &= ~kEsdfFixed;
          if (a.incremental) f |= kClsNb;
        }
        es &= 0xFFu;  // parent.setZero() (also after updateVoxelFromNeighbors set it, :197)
      } else {
        const bool efixed = (es & kEsdfFixed) != 0;
        if (tsdf_fixed || efixed) {
          if (!tsdf_fixed) {                                             // :211-220
            ed = sgn_default;
            es = (es & 0xFFu & ~kEsdfFixed) | kEsdfInQueue;
            f |= kClsRaise | kClsOpen;
          } else if ((ed > 0.0f && td + c.min_diff < ed) || (ed <= 0.0f && td - c.min_diff > ed)) {  // lower :221-237
            ed = td;
            es = (es & 0xFFu) | kEsdfFixed | kEsdfInQueue;
            f |= kClsOpen;
          } else if ((ed > 0.0f && td - c.min_diff > ed) || (ed <= 0.0f && td + c.min_diff < ed)) {  // raise :238-256
            ed = td;
            es = (es & 0xFFu) | kEsdfFixed | kEsdfInQueue;
            f |= kClsRaise | kClsOpen;
          }
        } else if (signum(td) != signum(ed)) {                           // :257-277
          if (td < ed) {
            ed = sgn_default;
            es = (es & 0xFFu) | kEsdfInQueue;
            f |= kClsOpen;
          } else {
            ed = sgn_default;
            es &= 0xFFu;
            f |= kClsRaise;
          }
        }
      }
      es |= kEsdfObserved;          // :282-283
      es &= ~kEsdfHallucinated;
    }
  }
  a.sh_d[i] = ed;
  a.sh_s[i] = es;
  a.sh_f[i] = f;
}

// updateVoxelFromNeighbors (:498-530) for the voxels that call it, against the walk's state at their moment.
// One workgroup per listed block.  A voxel depends on voxels IN FRONT of it in the walk only; inside a block those are the 13
// neighbours with a smaller linear index, and t = x + 2 y + 4 z is smaller for every one of them (vps <= 16 ... 32: the
// weights only have to exceed 1 and 1 + 2), so the block is swept in hyperplanes of equal t — thread (y, z) owns the one voxel
// of its row on the plane — with the block's shadow distances in LDS: one launch settles everything inside the blocks, and a
// launch is repeated only while values still cross block faces (chains along the walk are long: a row alternates between
// "took the estimate of its -x neighbour" and "that one is beyond max_distance now", and per-voxel Jacobi rounds needed 25 - 200).
// Round 6: everything a sweep reads that does NOT change during the sweep is staged in LDS first — the block's own voxels as the
// layer holds them (what a looker sees of a neighbour BEHIND it in the walk) and the one-voxel halo around the block (the 26
// neighbouring blocks: their shadow values if the walk has passed them, their layer values otherwise; fixed for the launch:
// other workgroups only write their shadows when they are done) as one padded (VPS + 2)^3 array of distances and one of
// "observed" bytes.  A plane then costs 26 LDS reads per looker instead of a chain of four dependent global loads (block table
// -> list position -> block flags -> voxel); the launch was 0.4 - 0.95 ms, four of them per update.
template <int VPS>
__global__ void __launch_bounds__(VPS * VPS) k_cls_nb_block(ClsArgs a) {
  constexpr int NV = VPS * VPS * VPS;
  constexpr int P = VPS + 2, NP = P * P * P;
  __shared__ float s_d[NV];        // the shadow distances of this sweep (what a looker sees of a neighbour IN FRONT of it)
  __shared__ uint8_t s_f[NV];      // shadow flags | 0x80: the shadow state is observed
  __shared__ float s_st[NP];       // static view: own voxels as the layer has them, halo voxels as described above
  __shared__ uint8_t s_so[NP];     // ... their "observed" bit (an absent voxel: 0)
  __shared__ uint32_t s_moved;
  const uint32_t p = blockIdx.x;
  const uint32_t slot = a.list_slots[p];
  if (!cls_valid(a, slot)) return;
  const int tid = threadIdx.x;
  const size_t base = (size_t)p * NV;
  // hyperplanes that hold a voxel which looks at its neighbours: the others are skipped (in a steady frame the lookers are
  // the new voxels along the frontier — a few planes per block)
  __shared__ uint8_t s_plane[7 * (VPS - 1) + 1];
  for (int i = tid; i <= 7 * (VPS - 1); i += VPS * VPS) s_plane[i] = 0;
  if (tid == 0) s_moved = 0;
  __syncthreads();
  // where the 27 neighbouring blocks' voxels are read from: kind (0 nothing there, 1 the shadow of list position idx, 2 the layer's
  // pool slot idx) — a chain of three dependent loads (block table -> list position -> block flags) done ONCE per neighbour block,
  // not once per halo voxel
  __shared__ uint32_t s_src[27];
  if (tid < 27) {
    uint32_t src = 0;
    const uint32_t s2 = tid == 13 ? slot : a.nb27[(size_t)p * 27 + tid];
    if (tid == 13) {
      src = 2u | (slot << 2);            // own block, behind the looker: the layer (zeros if it holds no ESDF voxels yet, :145)
    } else if (s2 != kInvalidSlot) {
      const uint32_t p2 = a.slot_pos[s2];
      const bool walked = p2 != kInvalidSlot && cls_valid(a, s2);   // the walk visits the neighbour's block ...
      const bool in_front = walked && p2 < p;                       // ... and has passed it
      // getVoxelPtrByGlobalIndex: the ESDF block exists if it did before the update or the walk has reached it (:145)
      if (in_front) src = 1u | (p2 << 2);
      else if (a.m.blk_flags[s2] & kFlagEsdfAlloc) src = 2u | (s2 << 2);
    }
    s_src[tid] = src;
  }
#pragma unroll 4
  for (int i = tid; i < NV; i += VPS * VPS) {
    const uint8_t f = a.sh_f[base + i];
    s_d[i] = a.sh_d[base + i];
    uint8_t sg = 0;   // a looker's TSDF sign (its ESDF value before the look is sign * default): 0x40 positive, 0x20 negative
    if (f & kClsNb) {
      const float td = a.m.dist[slot * NV + i];
      sg = td > 0.f ? 0x40u : (td < 0.f ? 0x20u : 0u);
      s_plane[(i % VPS) + 2 * ((i / VPS) % VPS) + 4 * (i / (VPS * VPS))] = 1;
    }
    s_f[i] = (uint8_t)((f & 0x1Fu) | sg | ((a.sh_s[base + i] & kEsdfObserved) ? 0x80u : 0u));
  }
  __syncthreads();
  // the static view, cell (px, py, pz) of the padded cube = voxel (px - 1, py - 1, pz - 1) relative to this block
#pragma unroll 4
  for (int i = tid; i < NP; i += VPS * VPS) {
    const int px = i % P, py = (i / P) % P, pz = i / (P * P);
    int nx = px - 1, ny = py - 1, nz = pz - 1;
    int cx = 1, cy = 1, cz = 1;
    if (nx < 0) { nx += VPS; cx = 0; } else if (nx >= VPS) { nx -= VPS; cx = 2; }
    if (ny < 0) { ny += VPS; cy = 0; } else if (ny >= VPS) { ny -= VPS; cy = 2; }
    if (nz < 0) { nz += VPS; cz = 0; } else if (nz >= VPS) { nz -= VPS; cz = 2; }
    const uint32_t nlin = (uint32_t)(nx + VPS * (ny + VPS * nz));
    const uint32_t src = s_src[cx + 3 * cy + 9 * cz];
    const size_t at = (size_t)(src >> 2) * NV + nlin;
    float nd = 0.f;
    uint32_t ns = 0u;
    if ((src & 3u) == 1u) {
      nd = a.sh_d[at];
      ns = a.sh_s[at];
    } else if ((src & 3u) == 2u) {
      nd = a.e.dist[at];
      ns = a.e.state[at];
    }
    s_st[i] = nd;
    s_so[i] = (ns & kEsdfObserved) ? 1 : 0;
  }
  __syncthreads();
  const EsdfCfgDev& c = a.c;
  const int y = tid % VPS, z = tid / VPS;
  uint32_t moved = 0;
  for (int t = 0; t <= 7 * (VPS - 1); ++t) {
    if (!s_plane[t]) continue;   // (uniform over the workgroup: no barrier is skipped by a part of it)
    const int x = t - 2 * y - 4 * z;
    if (x >= 0 && x < VPS) {
      const uint32_t lin = (uint32_t)(x + VPS * (y + VPS * z));
      const uint8_t f = (uint8_t)(s_f[lin] & 0x1Fu);
      if (f & kClsNb) {
        // the voxel as k_cls_base left it: sign * default, not fixed
        const float vd = (float)((s_f[lin] & 0x40u) ? 1 : ((s_f[lin] & 0x20u) ? -1 : 0)) * c.default_distance;
        float new_d = vd;
        bool hit = false;
        // the first neighbour in LUT order that qualifies (:505-527); an absent / unusable neighbour reads as not observed
#pragma unroll
        for (int idx = 0; idx < 26; ++idx) {
          const int nx = x + c_nb_off[idx][0], ny = y + c_nb_off[idx][1], nz = z + c_nb_off[idx][2];
          const bool inside = nx >= 0 && nx < VPS && ny >= 0 && ny < VPS && nz >= 0 && nz < VPS;
          const uint32_t nlin = (uint32_t)(nx + VPS * (ny + VPS * nz));
          float nd;
          bool obs;
          if (inside && nlin < lin) {   // same block, in front of the voxel: this sweep's shadow
            nd = s_d[nlin];
            obs = (s_f[nlin] & 0x80u) != 0;
          } else {
            const int pi = (nx + 1) + P * ((ny + 1) + P * (nz + 1));
            nd = s_st[pi];
            obs = s_so[pi] != 0;
          }
          if (hit || !obs || nd >= c.max_distance || nd <= -c.max_distance) continue;
          if (signum(nd) == signum(vd) && fabsf(nd) < fabsf(vd)) {
            new_d = nd + (float)signum(vd) * (idx < 6 ? 1.0f : (idx < 18 ? (float)1.4142135623730951 : (float)1.7320508075688772));   // NOT scaled by the voxel size (:508, :522)
            hit = true;
          }
        }
        // (the state word stays as k_cls_base left it: a hit sets in_queue when the voxel is committed, :187-190; the parent is zeroed right after, :197)
        const uint8_t f_new = hit ? (uint8_t)(f | kClsOpen | kClsHit) : (uint8_t)(f & ~(kClsOpen | kClsHit));
        if (__float_as_uint(s_d[lin]) != __float_as_uint(new_d) || f_new != f) {
          s_d[lin] = new_d;
          s_f[lin] = (uint8_t)(f_new | (s_f[lin] & 0xE0u));
          ++moved;
        }
      }
    }
    __syncthreads();
  }
  if (moved) atomicAdd(&s_moved, moved);
  __syncthreads();
  if (s_moved) {   // (a block that did not move leaves its shadow alone: readers of other blocks see the same bits)
    for (int i = tid; i < NV; i += VPS * VPS)
      if (s_f[i] & kClsNb) { a.sh_d[base + i] = s_d[i]; a.sh_f[base + i] = (uint8_t)(s_f[i] & 0x1Fu); }
    if (tid == 0) atomicAdd(&a.counters[1], s_moved);
  }
}

// shadow -> layer; which queue every push goes to; pushes per queue (one LDS histogram per workgroup)
__global__ void __launch_bounds__(256) k_cls_commit(ClsArgs a) {
  __shared__ uint32_t s_cnt[kStrictMaxBuckets + 2];
  for (int k = threadIdx.x; k <= a.num_buckets; k += 256) s_cnt[k] = 0;
  __syncthreads();
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)a.n_list * a.m.nvox) {
    const uint32_t p = (uint32_t)(i / a.m.nvox), lin = (uint32_t)(i % a.m.nvox);
    const uint32_t slot = a.list_slots[p];
    const uint8_t f = a.sh_f[i];
    if (lin == 0 && cls_valid(a, slot))   // :145-147 allocateBlockPtrByIndex + set_updated(true) (kMap only)
      atomicOr(&a.m.blk_flags[slot], kFlagEsdfAlloc | (1u << kFlagEsdfUpdShift) | kFlagEsdfDirty);
    if (f & kClsWrite) {
      const uint32_t gid = slot * a.m.nvox + lin;
      const float d = a.sh_d[i];
      a.e.dist[gid] = d;
      a.e.state[gid] = a.sh_s[i] | ((f & kClsHit) ? kEsdfInQueue : 0u);
      if (f & kClsOpen) {
        // BucketQueue::push (bucket_queue.h:41-50)
        double value = (double)d;
        const double max_val = (double)a.c.max_distance;
        if (value > max_val) value = max_val;
        int b = (int)floor(fabs(value) / max_val * (double)(a.num_buckets - 1));
        if (b >= a.num_buckets) b = a.num_buckets - 1;
        if (b < 0) b = 0;
        a.sh_q[i] = (uint8_t)b;
        atomicAdd(&s_cnt[b], 1u);
      }
      if (f & kClsRaise) atomicAdd(&s_cnt[a.num_buckets], 1u);
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k <= a.num_buckets; k += 256)
    if (s_cnt[k]) atomicAdd(&a.counters[8 + k], s_cnt[k]);
}

// arena chunks for every queue's entries; FIFO indices into the replay's control block (one thread)
__global__ void k_cls_reserve(ClsArgs a, rp::Args ra) {
  // (behind what addNewRobotPosition left in the queues: tails, reserved chunks and chunk_top are zero otherwise)
  rp::Ctl& c = *ra.ctl;
  uint32_t top = c.chunk_top;
  for (int q = 0; q <= a.num_buckets; ++q) {
    const uint32_t base = c.tail[q];
    const uint32_t n = base + a.counters[8 + q];
    const uint32_t chunks = (n + rp::kChunk - 1) / rp::kChunk;
    for (uint32_t j = c.reserved[q]; j < chunks; ++j) {
      if (top >= ra.max_chunks) { c.error |= 4u; break; }
      ra.chunk_tab[(size_t)q * ra.max_chunks + j] = top++;
    }
    a.counters[kClsBase + q] = base;
    c.head[q] = 0;
    c.tail[q] = n;
    if (chunks > c.reserved[q]) c.reserved[q] = chunks;
    c.k_cur[q] = 0;
  }
  c.chunk_top = top;
}

// the pushes of queues q0 .. q0 + rp::kScanC - 1 in walk order
struct ClsPushScan {
  const ClsArgs& a;
  const rp::Args& ra;
  int q0;
  __device__ uint32_t mask(uint32_t i) const {   // bit k: the item has an entry for queue q0 + k
    const uint8_t f = a.sh_f[i];
    uint32_t m = 0;
    if (!(f & kClsWrite)) return 0;
    if (f & kClsOpen) {
      const int k = (int)a.sh_q[i] - q0;
      if (k >= 0 && k < rp::kScanC) m |= 1u << k;
    }
    if (f & kClsRaise) {
      const int k = a.num_buckets - q0;
      if (k >= 0 && k < rp::kScanC) m |= 1u << k;
    }
    return m;
  }
  using State = uint32_t;   // (the item's mask: count() leaves it to apply())
  __device__ rp::Cnt4 count(uint32_t i, State& st) const {
    const uint32_t m = mask(i);
    st = m;
    rp::Cnt4 c{};
    for (int k = 0; k < rp::kScanC; ++k) c.v[k] = (m >> k) & 1u;
    return c;
  }
  __device__ void apply(uint32_t i, const rp::Cnt4& ex, const State& st) const {
    const uint32_t m = st;
    if (!m) return;
    const uint32_t gid = a.list_slots[i / a.m.nvox] * a.m.nvox + i % a.m.nvox;
    for (int k = 0; k < rp::kScanC; ++k)
      if ((m >> k) & 1u) rp::rp_queue_store(ra, q0 + k, a.counters[kClsBase + q0 + k] + ex.v[k], gid);
  }
};
__global__ void __launch_bounds__(kRpThreads) k_cls_push(ClsArgs a, rp::Args ra, RpScan sc, int q0) {
  __shared__ uint32_t s_last;
  // (nothing to do for these queues: leave the scan's ticket / generation alone)
  uint32_t any = 0;
  for (int k = 0; k < rp::kScanC && q0 + k <= a.num_buckets; ++k) any |= a.counters[8 + q0 + k];
  if (!any) return;
  ClsPushScan f{a, ra, q0};
  rp_scan_tiles(f, sc, a.n_list * a.m.nvox, a.counters + kClsScanTot, &ra.ctl->error);   // (the totals are not used: [8 + q] holds them already)
  // the last workgroup resets the ticket and moves the generation on for the next scan
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) s_last = (atomicAdd(&a.counters[3], 1u) == gridDim.x - 1) ? 1u : 0u;
  __syncthreads();
  if (s_last && threadIdx.x == 0) {
    a.counters[3] = 0;
    sc.ticket[0] = 0;
    sc.ticket[1] = sc.ticket[1] + 1;
  }
}

}  // namespace
