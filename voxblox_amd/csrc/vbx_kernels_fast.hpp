// vbx_kernels_fast.hpp — FastTsdfIntegrator: start-voxel set replay, voxel lists, early-termination solver, reference-set replay
// Part of libvbx_hip.so's single translation unit (included by vbx_hip.hip, in order).

namespace {
// ---------------------------------------------------------------------------
// kernels: Fast integrator (tsdf_integrator.cc:488-590)
// ---------------------------------------------------------------------------
// (the start-voxel keys are written by k_prep_points)

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
// Second half of the replay: the last probe of every slot leaves its hash in the set (separate launch
// so that no thread can read a slot after this frame has written it) — and, per point, the keep flag of
// the rays that survived the start-voxel test (compaction keeps visiting order).
__global__ void k_fast_start_commit_and_flags(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                              uint32_t n, uint32_t* set_vals, uint32_t offset,
                                              const uint8_t* __restrict__ flags, uint32_t* keep, DevState* st) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  keep[i] = (i < n && (flags[i] & 1)) ? 1u : 0u;  // indexed by visiting position s
  if (i >= n) return;
  const uint64_t key = keys[i];                   // indexed by sorted position
  if (key == ~0ull) return;
  const uint32_t slot = (uint32_t)(key >> 32);
  const bool last = (i + 1 >= n) || ((uint32_t)(keys[i + 1] >> 32) != slot);
  if (last) {
    set_vals[slot + offset] = vals[i];
    if (slot + offset == 0) st->sentinel_cleared = 1;
  }
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
#ifndef VBX_LIST_RPW
#define VBX_LIST_RPW 64
#endif
constexpr int kListRPW = VBX_LIST_RPW;  // rays per wave in k_fast_build_lists (32 and 16 measured: see HISTORY.md §4.3b)
template <int RPW>
__global__ void __launch_bounds__(256)
k_fast_build_lists(RayTab tab, CastCfg c, MapDev m, const uint32_t* __restrict__ off,
                   uint32_t* vox, uint32_t* vhash, uint32_t vox_cap, uint32_t* new_list, const uint32_t* __restrict__ redo_in,
                   uint32_t* redo_out, DevState* st, uint32_t* fix, uint32_t fix_cap) {
  // First pass (redo_in == nullptr): blocks met for the first time are inserted into the map
  // here (the block part of allocateStorageAndGetVoxelPtr, tsdf_integrator.cc:97-126); they
  // only get their pool slot after this kernel.  The list entries inside such a block are left
  // open and recorded as {list position, table position of the block, voxel} in `fix`;
  // k_fast_fixup fills them in once the slots are assigned.  (Until round 3 the rays that had
  // crossed a new block were walked a second time: a few hundred rays in steady state, but a
  // launch lasts as long as its longest walk — 85 us of a 1.4 ms frame.)  That second pass
  // (redo_in = the queued rays) remains for the frames whose open entries overflow `fix`: the
  // first frames of a map, where every block is new.
  if (redo_in && !st->fix_overflow) return;
  // RPW rays per wave, one per lane in the low lanes: the walk is a serial dependency chain per
  // ray, so with all 64 lanes busy the 70k rays of a frame are ~1 wave per SIMD and nothing
  // hides the latencies; fewer rays per wave means more resident waves.  Every ray lane stages
  // 16 list entries in LDS, then ALL 64 lanes write them out ray by ray as 64-byte runs (4 rays
  // per store instruction) instead of scattered dwords.
  __shared__ uint32_t s_buf[4][RPW][17];  // [wave][ray][entry], padded against bank conflicts
  __shared__ uint32_t s_h[4][RPW][17];    // the entries' LongIndexHash values (what the observed-set replay probes with)
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
    uint32_t pend = 0;  // 1 + table position of the current block while it has no slot yet
    bool redo = false;
    auto lookup = [&](uint64_t key) -> uint64_t {  // slot | (1 + table position) << 32 for a block without one
      const uint32_t sl = map_find(m, key);
      uint32_t pos1 = 0;
      if (sl == kInvalidSlot) {
        if (redo_in) {
          atomicOr(&st->error, 2u);
        } else {
          pos1 = map_insert_key_pos(m, key, new_list, st) + 1u;
          redo = true;
        }
      }
      return ((uint64_t)pos1 << 32) | sl;
    };
    for (uint32_t k0 = 0; __any(k0 < len); k0 += 16) {
      // (a) 16 DDA steps, no memory traffic: linear voxel index + "enters a new block" mark
      // per entry, the new blocks' keys on the side.  A hash lookup inside this loop would
      // stall the whole wave at almost every step (some lane always crosses a block face).
      uint64_t* tkeys = s_keys[wv][lane < RPW ? lane : 0];  // at most one block change per step
      int nt = 0;
      for (int j = 0; j < 16; ++j) {
        uint32_t e = 0xFFFFFFFFu;
        if (lane < RPW) s_h[wv][lane][j] = bw.h;
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
        if (t < nt) tkeys[t] = lookup(tkeys[t]);
      // open entries of this chunk (blocks without a slot yet): ONE reservation in `fix` per wave and chunk, and none
      // once the buffer has overflowed — a frame in which every block is new has millions of them, and a reservation
      // per step was 126 ms of same-address atomics at 0.02 m
      bool lane_pending = pend != 0;
      for (int t = 0; t < nt; ++t) lane_pending |= (tkeys[t] >> 32) != 0;
      uint32_t fix_at = 0xFFFFFFFFu;  // this lane's first item (0xFFFFFFFF: do not record)
      if (__any(lane_pending) && !redo_in) {
        uint32_t cnt = 0;
        if (lane_pending) {
          uint32_t sl = slot, pd = pend;
          for (int j = 0; j < 16; ++j) {
            const uint32_t e = (lane < RPW) ? s_buf[wv][lane][j] : 0xFFFFFFFFu;
            if (e == 0xFFFFFFFFu) continue;
            if (e & 0x80000000u) {
              const uint64_t tk = tkeys[(e >> 24) & 0x7Fu];
              sl = (uint32_t)tk;
              pd = (uint32_t)(tk >> 32);
            }
            cnt += (sl == kInvalidSlot && pd != 0) ? 1u : 0u;
          }
        }
        uint32_t incl = cnt;  // inclusive scan over the lanes
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t o = __shfl_up(incl, d);
          if (lane >= d) incl += o;
        }
        const uint32_t total = __shfl(incl, 63);
        if (total && !__hip_atomic_load(&st->fix_overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          uint32_t at = 0;
          if (lane == 0) at = atomicAdd(&st->fix_count, total);
          at = __shfl(at, 0);
          if (at + total <= fix_cap) fix_at = at + incl - cnt;
          else if (lane == 0) st->fix_overflow = 1;
        }
      }
      // (c) entries -> global voxel ids
      for (int j = 0; j < 16; ++j) {
        const uint32_t e = (lane < RPW) ? s_buf[wv][lane][j] : 0xFFFFFFFFu;
        uint32_t gid = 0xFFFFFFFFu;
        if (e != 0xFFFFFFFFu) {
          if (e & 0x80000000u) {
            const uint64_t tk = tkeys[(e >> 24) & 0x7Fu];
            slot = (uint32_t)tk;
            pend = (uint32_t)(tk >> 32);
          }
          if (slot != kInvalidSlot) {
            gid = slot * m.nvox + (e & 0xFFFFFFu);
          } else if (pend != 0 && fix_at != 0xFFFFFFFFu) {
            fix[3 * (size_t)fix_at] = base + k0 + j;
            fix[3 * (size_t)fix_at + 1] = pend - 1u;
            fix[3 * (size_t)fix_at + 2] = e & 0xFFFFFFu;
            ++fix_at;
          }
        }
        if (lane < RPW) s_buf[wv][lane][j] = gid;
      }
      // wave-synchronous flush (same wave wrote and reads; LDS ops of one wave are ordered)
      const int sub = lane >> 4, e = lane & 15;
      for (int q = 0; q < RPW / 4; ++q) {
        const int src = q * 4 + sub;  // lane whose ray is being written
        const uint32_t sbase = __shfl(base, src);
        const uint32_t slen = __shfl(len, src);
        if (k0 + e < slen) {
          vox[sbase + k0 + e] = s_buf[wv][src][e];
          vhash[sbase + k0 + e] = s_h[wv][src][e];
        }
      }
    }
    if (redo) redo_out[atomicAdd(&st->redo_count, 1u)] = r;
  }
}

// The open entries of k_fast_build_lists once k_assign_slots has run: entry = slot of its block * nvox + voxel.
__global__ void k_fast_fixup(MapDev m, const uint32_t* __restrict__ fix, uint32_t fix_cap, uint32_t* vox, DevState* st) {
  if (st->fix_overflow) return;  // the queued rays are rebuilt as a whole
  const uint32_t n = min(st->fix_count, fix_cap);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t slot = __hip_atomic_load(&m.hvals[fix[3 * (size_t)i + 1]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (slot == kInvalidSlot) {
      atomicOr(&st->error, 2u);  // (with error bit 0: the pool ran out and the host grows it)
      continue;
    }
    vox[fix[3 * (size_t)i]] = slot * m.nvox + fix[3 * (size_t)i + 2];
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
  int h_only;               // 1: only tighten the upper bounds from the certain claims and publish the possible claims
                            //    up to them (first sweep behind k_fast_seed_claims; the lower bounds stay)
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
  bool doneL = (len == 0) || a.init || a.h_only, done = (len == 0);
  if (a.init) tl = 0;
  if (a.h_only) tl = tl_old;
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
    const bool final_ray = !a.init && !a.l_only && !a.h_only && (tl == th) && (brokeL == brokeH);
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

// Start of the solve: every ray probes its first three voxels whatever the other rays do (the walk only stops on MORE
// than max_consecutive_ray_collisions collisions in a row, tsdf_integrator.cc:536-543), so those are certain claims and
// TL = min(len, max_consecutive + 1) a valid lower bound from the start.  The first sweep then bounds every ray from
// above with them and publishes possible claims only up to that bound — instead of along the whole path (4.3 M claims
// per 640x480 frame at 0.05 m, the most expensive kernel of the frame).  One thread per ray.
__global__ void k_fast_seed_claims(const uint32_t* __restrict__ off, const uint32_t* __restrict__ vox, uint32_t R, uint32_t* cl,
                                   uint32_t tag_cl, int s_bits, int max_consecutive, uint32_t* TL, uint32_t* TH, uint32_t* U) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r == R) {  // terminators of the exclusive scans over U and T
    U[R] = 0;
    TL[R] = 0;
  }
  if (r >= R) return;
  const uint32_t beg = off[r], len = off[r + 1] - beg;
  // (a negative max_consecutive_ray_collisions: the reference breaks at the first probe of every ray)
  const uint32_t t = max_consecutive < 0 ? min(len, 1u) : min(len, (uint32_t)max_consecutive + 1u);
  const uint32_t cl_val = (tag_cl << s_bits) | r;
  for (uint32_t k = 0; k < t; ++k) {
    const uint32_t gid = vox[beg + k];
    if (gid != 0xFFFFFFFFu) claim_min(&cl[gid], cl_val);
  }
  TL[r] = t;
  TH[r] = len;
}

// The solver stopped before every ray was final (reference observed set: the replay finishes the job): an open ray enters
// the replay with its upper bound.
__global__ void k_fast_open_guess(const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_dev, uint32_t n_bound,
                                  uint32_t* T, const uint32_t* __restrict__ TH) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= min(n_bound, *n_dev)) return;
  const uint32_t r = list[i];
  T[r] = TH[r];
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
    if (a.init) {  // terminators of the exclusive scans over U and T
      a.U[R] = 0;
      a.TL[R] = 0;
    }
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
__global__ void k_strict_keys(const uint32_t* __restrict__ poff, uint32_t r_begin, uint32_t r_end, uint32_t n_bound,
                              const uint32_t* __restrict__ off, const uint32_t* __restrict__ vhash, uint64_t* keys,
                              DevState* st) {
  // 16 lanes per ray of [r_begin, r_end): the ray's probes are written as one run.  The number of probes
  // is only known on the device (rounds run in batches without a host check): the host sized the key
  // buffers for n_bound; a round that needs more raises rp_overflow and the rest of the batch idles.
  // The hashes come from the list k_fast_build_lists left beside the voxel ids; four loads per lane are in flight
  // (a ray of 400 probes is otherwise a chain of 25 dependent trips that the whole launch waits for).
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t p_begin = poff[r_begin], P = poff[r_end] - p_begin;
  if (t == 0) {
    st->rp_n = P;
    if (P > n_bound) st->rp_overflow = 1;
  }
  if (P > n_bound || st->rp_overflow) return;
  const uint32_t r = r_begin + (t >> 4);
  if (r >= r_end) return;
  const uint32_t p0 = poff[r], n = poff[r + 1] - p0, beg = off[r];
  for (uint32_t k = t & 15u; k < n; k += 64) {
    uint32_t h[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = vhash[beg + ((k + 16 * j < n) ? k + 16 * j : k)];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t kk = k + 16 * j;
      if (kk >= n) break;
      const uint32_t p = p0 + kk;  // ascends in (ray, step) order = time
      keys[p - p_begin] = ((uint64_t)(h[j] & 0xFFFFFu) << 44) | ((uint64_t)(h[j] >> 20) << 32) | p;
    }
  }
}
__device__ inline uint32_t strict_key_slot(uint64_t key) { return (uint32_t)(key >> 44); }
__device__ inline uint32_t strict_key_hash(uint64_t key) {
  return (uint32_t)(key >> 44) | ((uint32_t)((key >> 32) & 0xFFFu) << 20);
}
// replaceHash outcome of every probe (approx_hash_array.h:125-134): collision = the slot held
// this hash already.  set_vals = pseudo_set_ as the frame found it (at offset_).
__global__ void k_strict_outcome(const uint64_t* __restrict__ keys, const DevState* __restrict__ st,
                                 const uint32_t* __restrict__ set_vals, uint32_t offset, int sentinel_live,
                                 uint8_t* collided_by_p) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t P = st->rp_n;
  if (st->rp_overflow || i >= P) return;
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
k_strict_scan(const uint32_t* __restrict__ poff, const uint32_t* __restrict__ off, uint32_t R, uint32_t r_begin,
              uint32_t r_end, const uint8_t* __restrict__ collided, int max_consecutive, uint32_t* T, uint32_t* U,
              uint8_t* moved, uint32_t round_idx, uint32_t grow_mult, DevState* st) {
  // Only the rays of [r_begin, r_end) are in play: the grid covers that range and T is updated IN PLACE (a ray reads
  // nothing but its own count and its own outcomes).  Rays outside keep their count (they are final, or not in play
  // yet); an idling round (rp_overflow) changes nothing.
  // 16 lanes per ray: 16 outcomes per step, the consecutive-collision counter is the run length
  // of the ballot mask (as in sweep_ray); the next step's outcomes are in flight while this step's ballots run
  constexpr int G = 16;
  const int lane = threadIdx.x & 63;
  const int grp = lane / G, gl = lane % G;
  const uint32_t r = r_begin + (blockIdx.x * blockDim.x + threadIdx.x) / G;
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // terminators of the exclusive scans over T and U
    T[R] = 0;
    U[R] = 0;
  }
  const bool ray_ok = r < r_end && !st->rp_overflow;
  const uint32_t t = ray_ok ? T[r] : 0;
  const uint32_t len = ray_ok ? off[r + 1] - off[r] : 0;
  const uint32_t p0 = ray_ok ? poff[r] : 0;
  const unsigned long long gmask = (1ull << G) - 1ull;
  const unsigned long long below = (2ull << gl) - 1ull;
  int carry = 0;
  uint32_t tn = t;
  bool broke = false, done = (t == 0);
  uint8_t c_pf = (gl < t) ? collided[p0 + gl] : 0;
  for (uint32_t base = 0; __any(!done); base += G) {
    const uint32_t k = base + gl;
    const bool act = !done && k < t;
    const bool c = act && c_pf != 0;
    c_pf = (!done && k + G < t) ? collided[p0 + k + G] : 0;
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
    if (!broke && t < len) tn = min(len, max(grow_mult * t, t + 16u));  // surplus probes vanish again next round
    if (tn != t) T[r] = tn;
    U[r] = broke ? tn - 1 : tn;  // the terminating probe's voxel is not updated
    moved[r] = (tn != t) ? 1 : 0;
    if (tn != t) st->rp_changed_round = round_idx + 1;  // same value from every writer of the round
  }
}
// Lowest ray index whose probe count moved in the last round: every ray below it is final.
__global__ void k_strict_first_moved(const uint8_t* __restrict__ moved, uint32_t R, uint32_t* out) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  const bool hit = r < R && moved[r];
  const unsigned long long b = __ballot(hit);
  if (b && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)b) - 1)) atomicMin(out, r);
}

// The last probe of every slot leaves its hash in the persistent set.
__global__ void k_strict_commit(const uint64_t* __restrict__ keys, uint32_t* set_vals, uint32_t offset,
                                DevState* st) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t P = st->rp_n;
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
  // new blocks join the Layer here (with their first-touch rank); block->updated().set() (tsdf_integrator.cc:128) for every
  // touched block is done by the fold, one thread per block
  const uint32_t slot = gid / m.nvox;
  const bool first_of_block = (k == 0) || (vox[off_full[r] + k - 1] / m.nvox != slot);
  if (first_of_block) publish_new_block_ranked(m, slot, st, ((unsigned long long)r << 24) | (unsigned long long)(k & 0xFFFFFFu));
}



}  // namespace

