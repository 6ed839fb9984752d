// vbx_kernels_esdf_replay.hpp — device side of the parallel reference-order open set (vbx_esdf_replay_core.hpp)
// Part of libvbx_hip.so's single translation unit (included by vbx_hip.hip, in order).
//
// k_rp_step runs ONE phase of the replay over the whole grid and lets the last workgroup to finish run rp_control,
// which writes the next phase into the control block.  The host launches the same kernel back to back and looks at
// Ctl::done now and then: a kernel boundary is the grid-wide barrier
// (1.5 - 2 us, MI355X_MICROARCH.md "boundary"), every phase gets the whole chip, no workgroup ever waits for
// another inside a launch except in the two SCAN phases, where a tile waits for tiles with smaller tickets only
// (chained scan with decoupled look-back: those tiles are running or done, so the wait ends).

#define RP_FN __device__ inline
// returning increment of a counter shared by every lane that gets here together: one atomic per wave (a single address
// takes ~90 atomics per microsecond; a PLACE phase asks for tens of thousands of target / dirty-list slots)
__device__ inline uint32_t rp_wave_inc(uint32_t* ctr) {
  const unsigned long long mask = __ballot(1);
  const int lane = (int)(threadIdx.x & 63);
  const int leader = __ffsll((long long)mask) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(ctr, (uint32_t)__popcll(mask));
  base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
  return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}
#define RP_INC(p) rp_wave_inc(p)
// PH_APPLY collects its dirty marks per workgroup (k_rp_step files them with one atomic on Ctl::n_dirty): defined below the
// LDS layout
__device__ inline bool rp_wg_dirty_push(uint32_t t);
#define RP_WG_DIRTY_PUSH(t) rp_wg_dirty_push(t)
// (the control functions' reads of counters that phases update: on the device rp_control runs on the LDS copy k_rp_step has just made
// with coherent reads — or, k_rp_begin, before anything else runs —, so these are plain reads; as LDS atomics they were a dozen
// to two dozen dependent round trips per control step)
#define RP_LD(x) (x)
#define RP_SHARD (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6))   // the wave's number: which of a sharded counter's lines it uses
#define RP_LD64(x) (x)
#define RP_LD_RO(x) __hip_atomic_load(&(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)   // a coherent READ (no read-modify-write)
// PH_APPLY finds an item's shard of the changed / born lists from Ctl::chg_pre / born_pre: a copy in LDS for the phase (g_rp_pre,
// filled by k_rp_step before the phase runs): [0 .. kShards] changed, [kShards + 1 ..] born
__device__ inline uint32_t rp_pre_lds(const uint32_t* arr, uint32_t k);
#define RP_PRE(arr, k) rp_pre_lds(arr, k)
#include "vbx_esdf_replay_core.hpp"

namespace {

// broadcast of lane `src` (wave-uniform): v_readlane instead of a trip through the LDS crossbar
__device__ inline uint32_t rl_u32(uint32_t x, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)x, __builtin_amdgcn_readfirstlane(src)); }
__device__ inline float rl_f32(float x, int src) { return __uint_as_float(rl_u32(__float_as_uint(x), src)); }
__device__ inline unsigned long long rl_u64(unsigned long long x, int src) {
  return (unsigned long long)rl_u32((uint32_t)x, src) | ((unsigned long long)rl_u32((uint32_t)(x >> 32), src) << 32);
}

__device__ inline void rp_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// inclusive prefix sum / prefix maximum over the 64 lanes on the DPP data path (values >= 0: lanes without a source read 0)
__device__ inline uint32_t rp_wave_scan_add(uint32_t x) {
  int v = (int)x;
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return (uint32_t)v;
}
__device__ inline uint32_t rp_wave_scan_max(uint32_t x) {
  int v = (int)x;
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));
  return (uint32_t)v;
}

constexpr int kRpThreads = 256;
constexpr uint32_t kWgDirtyCap = 4000;   // dirty marks a workgroup collects in PH_APPLY before it falls back to the shared counter
constexpr uint32_t kRpGraphSteps = 64;   // launches per batch (a power of two; Ctl::hdr numbers the launches modulo it)
constexpr uint32_t kRpSpinMax = 1u << 22;

struct RpScan {
  unsigned long long* desc;  // [tiles][4]: (generation << 2 | state) << 32 | value ; state 1 = tile aggregate, 2 = inclusive prefix
  uint32_t* ticket;          // [0] next tile, [1] generation (bumped by the last workgroup after every SCAN phase)
  uint32_t max_tiles;
};

// pool slot of every ESDF block's 27 neighbours (kNone where the ESDF layer has no block)
__global__ void k_rp_nbslot(MapDev m, uint32_t used, uint32_t* __restrict__ nbslot) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= used * 27u) return;
  const uint32_t slot = i / 27u, k = i % 27u;
  uint32_t out = rp::kNone;
  const uint32_t f = m.blk_flags[slot];
  if (!(f & kFlagFree) && (f & kFlagEsdfAlloc)) {
    const int dx = (int)(k % 3) - 1, dy = (int)(k / 3 % 3) - 1, dz = (int)(k / 9) - 1;
    const uint32_t s2 = map_find(m, pack_block_key(m.blk_idx[3 * slot] + dx, m.blk_idx[3 * slot + 1] + dy, m.blk_idx[3 * slot + 2] + dz));
    if (s2 != kInvalidSlot && (m.blk_flags[s2] & kFlagEsdfAlloc)) out = s2;
  }
  nbslot[i] = out;
}

// Args::hazard: does the voxel have an observed neighbour of the other sign class (d > 0 or not)?  Classes are fixed for the
// whole of processRaiseSet / processOpenSet, so this is computed once per update, after the voxel loop of updateFromTsdfBlocks.
__global__ void k_rp_hazard(rp::Args a, const uint32_t* __restrict__ blk_flags, uint32_t used, uint8_t* __restrict__ hazard) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)used * a.nvox) return;
  const uint32_t f = blk_flags[gid / a.nvox];
  uint8_t h = 0;
  if (!(f & kFlagFree) && (f & kFlagEsdfAlloc)) {
    const bool pos = a.dist[gid] > 0;
    for (int k = 0; k < 26; ++k) {
      const uint32_t n = rp::rp_neighbour(a, (uint32_t)gid, k);
      if (n == rp::kNone) continue;
      if ((a.state[n] & rp::kObserved) && ((a.dist[n] > 0) != pos)) { h = 1; break; }
    }
  }
  hazard[gid] = h;
}

template <bool SERIAL>
__device__ inline void rp_run_phase(const rp::Args& a, uint32_t phase, uint32_t tid) {
  if (SERIAL) {
    // (debug form, VBX_RP_SERIAL=1: the thread-per-item folds and ranking the CPU emulation runs)
    switch (phase) {
      case rp::PH_FOLD: rp::rp_phase_fold(a, tid); return;
      case rp::PH_SIM: rp::rp_phase_sim(a, tid); return;
      case rp::PH_COMMIT_FOLD: rp::rp_phase_commit_fold(a, tid); return;
      case rp::PH_RAISE_FOLD: rp::rp_phase_raise_fold(a, tid); return;
      default: break;
    }
  }
  switch (phase) {
    case rp::PH_PLACE_BASE: rp::rp_phase_place_base(a, tid); break;
    case rp::PH_APPLY: rp::rp_phase_apply(a, tid); break;
    case rp::PH_MINCUT: rp::rp_phase_mincut(a, tid); break;
    case rp::PH_RANK_WRITE: rp::rp_phase_rank_write(a, tid); break;
    case rp::PH_CLEANUP: rp::rp_phase_cleanup(a, tid); break;
    default: break;
  }
}

// rp_fold (vbx_esdf_replay_core.hpp, the form the CPU emulation runs) as ONE WAVE per target: a lane holds up to kEvQ of
// the target's events (only the slots a target's list reaches are looked at: most lists have fewer than 64 entries) with their
// records' pop times and pop-time states; the order of the events is a rank computed by comparing pop times across lanes.
//
// The replay itself (round 6).  Until round 5 it ran event by event, wave-uniformly, on broadcast values: a target of a crowded
// neighbourhood (a list of 256 events) cost 256 trips through ~100 wave instructions, and a launch lasts as long as its slowest
// target (first update of a map: 6,785 fold launches of 56 us).  But most events change nothing: an offer that does not beat the
// voxel's distance leaves no trace.  So the events are put into pop order through the wave's LDS scratch (rank r -> lane r % 64,
// slot r / 64) and taken 64 at a time: EVERY lane tests its own event against the voxel's current state (rp_relax with the
// lane's own LUT index), the first lane whose event changes the state is found with a ballot, that one event is applied
// wave-uniformly — the events in front of it were no-ops against exactly this state, so skipping them is the sequential
// result — and the test is repeated from the event behind it.  A pop of the voxel itself (it records the state it finds)
// always counts as a change.  Steps per target = events that change something + one per 64 events, instead of events.
// (commit: the pushes per queue go to push_acc — the workgroup's LDS counters in k_rp_step —, the relaxations to *relax_acc)
constexpr uint32_t kFoldLdsWords = rp::kEvMax * 4;   // per wave: {code, distance bits, state, pop-time state moved} per event
__device__ inline void rp_fold_wave(const rp::Args& a, uint32_t t, unsigned long long limit, bool commit, int lane, uint32_t* ws,
                                    uint32_t* push_acc = nullptr, uint32_t* relax_acc = nullptr) {
  using namespace rp;
  constexpr int Q = (int)kEvQ;     // events per lane: event e of the target lives in lane e % 64, slot e / 64
  Ctl& c = *a.ctl;
  const unsigned long long fk0 = wall_clock64();
  const uint32_t gid = a.tgt_gid[t];
  if (gid == kNone) return;
  const int b = (int)c.bucket;
  uint32_t n_all = a.tgt_cnt[t];
  if (n_all > a.c.ev) n_all = a.c.ev;
  const int slots = (int)((n_all + 63u) >> 6);   // slots in use (wave-uniform)
  const float d0 = a.dist[gid];          // (issued with the event loads, used after the ranking)
  const uint32_t s0 = a.state[gid];
  uint32_t code[Q], es[Q], emeta[Q];
  unsigned long long eT[Q];
  float ed[Q];
  bool valid[Q], have[Q], epoison[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    code[q] = 0; es[q] = 0; emeta[q] = 0; eT[q] = kNever; ed[q] = 0.f;
    valid[q] = false; have[q] = false; epoison[q] = false;
    const uint32_t e = (uint32_t)lane + 64u * q;
    if (e < n_all) {
      have[q] = true;
      code[q] = a.tgt_ev[(size_t)t * a.c.ev + e];
      const uint32_t r = code[q] >> 5;
      emeta[q] = a.rec_meta[r];
      eT[q] = a.rec_T[r];
      epoison[q] = a.rec_poison[r] != 0u;
      ed[q] = a.rec_d[r];
      es[q] = a.rec_s[r];
      valid[q] = rp_meta_live(emeta[q]) && !epoison[q] && eT[q] < limit;
    }
  }
  const unsigned long long fk1 = wall_clock64();
  // rank of every valid event = valid events with a smaller pop time (pop times of valid events are distinct)
  uint32_t rank[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) rank[q] = 0;
  uint32_t n = 0;
#pragma unroll
  for (int q2 = 0; q2 < Q; ++q2) {
    if (q2 >= slots) break;
    unsigned long long vm = __ballot(valid[q2]);
    n += (uint32_t)__popcll(vm);
    while (vm) {
      const int k = __ffsll((long long)vm) - 1;
      vm &= vm - 1;
      const unsigned long long Tk = rl_u64(eT[q2], k);
#pragma unroll
      for (int q = 0; q < Q; ++q) rank[q] += (Tk < eT[q]) ? 1u : 0u;
    }
  }
  // the events in pop order: rank r -> ws[4 r ..] (the wave's own LDS scratch; DS operations of one wave execute in order)
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    if (q < slots && valid[q]) {
      uint32_t* w = ws + 4u * rank[q];
      w[0] = code[q]; w[1] = __float_as_uint(ed[q]); w[2] = es[q]; w[3] = 0u;
    }
  }
  rp_wave_sync();
  const unsigned long long fk2 = wall_clock64();
  float d = d0;
  uint32_t s = s0;
  const bool usable = (s0 & kObserved) && !(s0 & kFixed);
  uint32_t relax = 0;
  // pushes below b seen in this fold: the j-th lives in lane j
  uint32_t lp_rec = kNone, lp_lb = 0, lp_s = 0;
  float lp_d = 0.f;
  uint32_t n_lp = 0;
  for (uint32_t base = 0; base < n; base += 64u) {
    const uint32_t cnt = n - base < 64u ? n - base : 64u;
    uint32_t mcode = 0, ms = 0;
    float md = 0.f;
    if ((uint32_t)lane < cnt) {
      const uint32_t* w = ws + 4u * (base + (uint32_t)lane);
      mcode = w[0]; md = __uint_as_float(w[1]); ms = w[2];
    }
    const uint32_t mlut = mcode & 31u;
    const bool is_pop = mlut == kOwn;
    // what does not depend on the voxel's state: the pop offers nothing (:389-392), the voxel cannot be written (:414-417)
    const bool dead = !is_pop && (!(ms & kObserved) || md >= a.c.max_distance || md <= -a.c.max_distance || !usable);
    uint32_t cur = 0;
    for (;;) {
      bool chg = false;
      float nd = 0.f;
      uint32_t np = 0;
      if ((uint32_t)lane >= cur && (uint32_t)lane < cnt) {
        if (is_pop) chg = true;
        else if (!dead) chg = rp_relax(a.c, md, ms, d, (int)mlut, &nd, &np);
      }
      const unsigned long long cm = __ballot(chg);
      if (!cm) break;
      const int f = __ffsll((long long)cm) - 1;   // events cur .. f - 1 leave the state as it is
      cur = (uint32_t)f + 1u;
      const uint32_t fcode = rl_u32(mcode, f);
      const uint32_t r = fcode >> 5, lut = fcode & 31u;
      if (lut == kOwn) {
        // the pop: processOpenSet reads the voxel here (:381-392)
        if (!commit) {
          const float evd = rl_f32(md, f);
          const uint32_t evs = rl_u32(ms, f);
          if (lane == 0) {
            a.rec_d_n[r] = d;
            a.rec_s_n[r] = s;
            // did the record's pop-time state move? (read back by the lane that holds the event)
            if (__float_as_uint(d) != __float_as_uint(evd) || s != evs) ws[4u * (base + (uint32_t)f) + 3u] = 1u;
          }
        }
        s &= ~kInQueue;                                            // :386
        continue;
      }
      const float fnd = rl_f32(nd, f);
      const uint32_t fnp = rl_u32(np, f);
      ++relax;
      d = fnd;
      s = (s & 0xFFu) | fnp;
      if (a.c.multi_queue || !(s & kInQueue)) {
        s |= kInQueue;
        const int nb = rp_bucket_of(a.c, fnd);
        if (commit) {
          if (lane == 0) {
            const uint32_t w = r * 7 + lut / 4, sh = (lut % 4) * 8;
            atomicOr(&a.rec_push[w], (uint32_t)(nb + 1) << sh);
            atomicAdd(push_acc ? &push_acc[nb] : &c.push_cnt[nb], 1u);
          }
        } else if (nb < b) {
          if (n_lp == 64) {
            // no room to describe this push: the pushing record leaves the super-step (the cut falls in front of it)
            if (lane == 0 && atomicExch(&a.rec_poison[r], 1u) == 0u) {
              atomicAdd(&c.st_poison, 1ull);
              if (r < c.K) atomicMin(&c.k_limit, r);
              const uint32_t base_r = a.rec_base[r];
              const unsigned long long Tr = a.rec_T[r];
              if (Tr != kNever && (Tr & kRankMask) != 0) atomicMin(&a.sub_restart[base_r], (uint32_t)(Tr & kRankMask) - 1u);
              if (atomicExch(&a.sub_dirty[base_r], 1u) == 0u) a.sd_list[atomicAdd(&c.n_sd, 1u)] = base_r;
              rp_chg_push(a, r);
            }
            continue;
          }
          if ((uint32_t)lane == n_lp) { lp_rec = r; lp_lb = lut | ((uint32_t)nb << 8); lp_d = d; lp_s = s; }
          ++n_lp;
        }
      }
    }
  }
  if (commit) {
    if (lane == 0) {
      if (d != d0 || s != s0) {
        a.dist[gid] = d;
        a.state[gid] = s;
        rp::rp_mark_block(a, gid);
      }
      if (relax) {
        if (relax_acc) *relax_acc += relax;
        else atomicAdd(&c.st_relax, (unsigned long long)relax);
      }
    }
    return;
  }
  rp_wave_sync();   // (lane 0's moved flags)
  const unsigned long long fk3 = wall_clock64();
  bool pop_moved[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) pop_moved[q] = (q < slots && valid[q] && (code[q] & 31u) == kOwn) ? ws[4u * rank[q] + 3u] != 0u : false;
  // ---- outputs of an iteration: the records on this voxel (their own-pop events, dead or alive)
  bool own[Q], chg[Q], found[Q];
  uint32_t mn[Q], pusher[Q], fbucket[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    own[q] = have[q] && (code[q] & 31) == kOwn && !epoison[q];
    chg[q] = false; found[q] = false;
    mn[q] = emeta[q];
    pusher[q] = kNone;
    fbucket[q] = rp_meta_bucket(emeta[q]);
    if (own[q]) pusher[q] = a.rec_pusher[code[q] >> 5];
  }
  unsigned long long matched = 0;  // push j was matched by a record of mine
  for (uint32_t j = 0; j < n_lp; ++j) {
    const uint32_t lr = rl_u32(lp_rec, (int)j), llb = rl_u32(lp_lb, (int)j);
#pragma unroll
    for (int q = 0; q < Q; ++q)
      if (own[q] && pusher[q] != kNone && lr == pusher[q] && (llb & 0xFF) == rp_meta_lut(emeta[q])) {
        found[q] = true; fbucket[q] = llb >> 8; matched |= 1ull << j;
      }
  }
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    if (!own[q]) continue;
    const uint32_t r = code[q] >> 5;
    if (pusher[q] != kNone) {
      // an excursion record lives iff this fold pushed it below b
      mn[q] = rp_meta(rp_meta_lut(emeta[q]), fbucket[q], found[q]) | (emeta[q] & (1u << 18));
      if (mn[q] != emeta[q]) chg[q] = true;
    }
    if (rp_meta_live(emeta[q]) && eT[q] != kNever) {
      if (pop_moved[q]) chg[q] = true;   // it popped in this fold (lane 0 wrote its pop-time state)
    } else {
      a.rec_d_n[r] = ed[q];
      a.rec_s_n[r] = es[q];
    }
    a.rec_meta_n[r] = mn[q];
    if (chg[q]) rp_chg_push(a, r);
  }
  // pushes below b that no record stands for yet
  unsigned long long any_matched = 0;
  for (uint32_t j = 0; j < n_lp; ++j)
    if (__ballot((matched >> j) & 1ull)) any_matched |= 1ull << j;
  if ((uint32_t)lane < n_lp && !((any_matched >> lane) & 1ull)) {
    if (a.rec_kid[(size_t)lp_rec * 26 + (lp_lb & 0xFF)] == 0u) {
      uint32_t* bw = rp_born_slot(a, lp_rec);   // (null: no room even to note it — the cut falls in front of its pusher)
      if (bw) {
        bw[0] = lp_rec;
        bw[1] = lp_lb & 0xFF;
        bw[2] = lp_lb >> 8;
        bw[3] = gid;
        bw[4] = __float_as_uint(lp_d);
        bw[5] = lp_s;
      }
    }
  }
  if (a.wg_stats && lane == 0 && blockIdx.x < 4096u) {   // (VBX_RP_STATS) what the folds of this wave cost
    unsigned long long* w = a.wg_stats + (size_t)4096 * rp::kWgStats + ((size_t)blockIdx.x * (kRpThreads / 64) + (threadIdx.x >> 6)) * 8;
    w[0] += 1; w[1] += n_all;
    w[2] += fk1 - fk0; w[3] += fk2 - fk1; w[4] += fk3 - fk2; w[5] += wall_clock64() - fk3;
    w[6] += n_all >= 64u ? 1 : 0; w[7] += n_all >= 192u ? 1 : 0;
  }
}

// rp_fold_wave for TWO targets at once, a half-wave each (round 6, last session).  A launch over more than 4,096 targets gives
// every wave several folds in a row, and a fold is latency — five microseconds of dependent trips and LDS round trips with a
// handful of events in 64 lanes (a steady update's lists hold six on average) — so such a launch lasts folds-per-wave times
// that.  Lists of up to 32 events are folded in pairs: lanes 0-31 hold the events of one target, lanes 32-63 of the other, one
// event per lane; ballots are taken apart per half, broadcasts go through ds_bpermute (the source lane is always a lane of the
// reader's own half, and a half's lanes leave every loop together: its trip counts are half-uniform), each half has half of the
// wave's scratch.  Same steps, same results; a pair with a longer list goes through rp_fold_wave one target after the other.
// (act: this half has a target; t, gid, n_all: half-uniform)
__device__ inline uint32_t rp_hshfl(uint32_t x, uint32_t src_lane) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)x); }
template <int G>   // lanes per target: 32 (two targets per wave) or 16 (four)
__device__ inline void rp_fold_pair(const rp::Args& a, uint32_t t, uint32_t gid, uint32_t n_all, bool act, unsigned long long limit, bool commit, int lane,
                                    uint32_t* ws_wave, uint32_t* push_acc, uint32_t* relax_acc) {
  using namespace rp;
  Ctl& c = *a.ctl;
  constexpr uint32_t kGm = G == 32 ? 0xFFFFFFFFu : (1u << (G & 31)) - 1u;
  const uint32_t h = (uint32_t)lane / G, hl = (uint32_t)lane % G, hb = h * G;
  uint32_t* ws = ws_wave + h * (kFoldLdsWords / (64 / G));
  act = act && gid != kNone;
  if (!act) n_all = 0;
  const int b = (int)c.bucket;
  const float d0 = act ? a.dist[gid] : 0.f;
  const uint32_t s0 = act ? a.state[gid] : 0u;
  uint32_t code = 0, es = 0, emeta = 0;
  unsigned long long eT = kNever;
  float ed = 0.f;
  bool valid = false, epoison = false;
  const bool have = hl < n_all;
  if (have) {
    code = a.tgt_ev[(size_t)t * a.c.ev + hl];
    const uint32_t r = code >> 5;
    emeta = a.rec_meta[r];
    eT = a.rec_T[r];
    epoison = a.rec_poison[r] != 0u;
    ed = a.rec_d[r];
    es = a.rec_s[r];
    valid = rp_meta_live(emeta) && !epoison && eT < limit;
  }
  // rank among the half's valid events (pop times of valid events are distinct; (base record, rank) fits 32 bits: a rank stays below kSimMax)
  const uint32_t key = (uint32_t)((eT >> kRankBits) << 11) | (uint32_t)(eT & 0x7FFull);
  const uint32_t vmask = (uint32_t)(__ballot(valid) >> hb) & kGm;
  const uint32_t n = (uint32_t)__popc(vmask);
  uint32_t rank = 0;
  for (uint32_t m = vmask; m; m &= m - 1u) {
    const uint32_t k = (uint32_t)__ffs((int)m) - 1u;
    rank += rp_hshfl(key, hb + k) < key ? 1u : 0u;
  }
  if (valid) {
    uint32_t* w = ws + 4u * rank;
    w[0] = code; w[1] = __float_as_uint(ed); w[2] = es; w[3] = 0u;
  }
  rp_wave_sync();
  float d = d0;
  uint32_t s = s0;
  const bool usable = (s0 & kObserved) && !(s0 & kFixed);
  uint32_t relax = 0;
  uint32_t lp_rec = kNone, lp_lb = 0, lp_s = 0;   // pushes below b seen in this fold: the j-th lives in lane j of the half
  float lp_d = 0.f;
  uint32_t n_lp = 0;
  {
    uint32_t mcode = 0, ms = 0;
    float md = 0.f;
    if (hl < n) {
      const uint32_t* w = ws + 4u * hl;
      mcode = w[0]; md = __uint_as_float(w[1]); ms = w[2];
    }
    const uint32_t mlut = mcode & 31u;
    const bool is_pop = mlut == kOwn;
    const bool dead = !is_pop && (!(ms & kObserved) || md >= a.c.max_distance || md <= -a.c.max_distance || !usable);
    uint32_t cur = 0;
    for (;;) {
      bool chg = false;
      float nd = 0.f;
      uint32_t np = 0;
      if (hl >= cur && hl < n) {
        if (is_pop) chg = true;
        else if (!dead) chg = rp_relax(a.c, md, ms, d, (int)mlut, &nd, &np);
      }
      const uint32_t cm = (uint32_t)(__ballot(chg) >> hb) & kGm;
      if (!cm) break;
      const uint32_t f = (uint32_t)__ffs((int)cm) - 1u;   // events cur .. f - 1 leave the state as it is
      cur = f + 1u;
      const uint32_t fcode = rp_hshfl(mcode, hb + f);
      const uint32_t fndb = rp_hshfl(__float_as_uint(nd), hb + f), fnp = rp_hshfl(np, hb + f);
      const uint32_t evdb = rp_hshfl(__float_as_uint(md), hb + f), evs = rp_hshfl(ms, hb + f);
      const uint32_t r = fcode >> 5, lut = fcode & 31u;
      if (lut == kOwn) {
        if (!commit && hl == 0) {
          a.rec_d_n[r] = d;
          a.rec_s_n[r] = s;
          if (__float_as_uint(d) != evdb || s != evs) ws[4u * f + 3u] = 1u;
        }
        s &= ~kInQueue;
        continue;
      }
      const float fnd = __uint_as_float(fndb);
      ++relax;
      d = fnd;
      s = (s & 0xFFu) | fnp;
      if (a.c.multi_queue || !(s & kInQueue)) {
        s |= kInQueue;
        const int nb = rp_bucket_of(a.c, fnd);
        if (commit) {
          if (hl == 0) {
            const uint32_t w = r * 7 + lut / 4, sh = (lut % 4) * 8;
            atomicOr(&a.rec_push[w], (uint32_t)(nb + 1) << sh);
            atomicAdd(push_acc ? &push_acc[nb] : &c.push_cnt[nb], 1u);
          }
        } else if (nb < b) {
          // (a list of at most G events makes at most G pushes: there is a lane for every one of them)
          if (hl == n_lp) { lp_rec = r; lp_lb = lut | ((uint32_t)nb << 8); lp_d = d; lp_s = s; }
          ++n_lp;
        }
      }
    }
  }
  if (commit) {
    if (hl == 0 && act) {
      if (d != d0 || s != s0) {
        a.dist[gid] = d;
        a.state[gid] = s;
        rp::rp_mark_block(a, gid);
      }
      if (relax) {
        if (relax_acc) *relax_acc += relax;
        else atomicAdd(&c.st_relax, (unsigned long long)relax);
      }
    }
    return;
  }
  rp_wave_sync();   // (lane 0's moved flags)
  const bool own = have && (code & 31u) == kOwn && !epoison;
  const bool pop_moved = (valid && (code & 31u) == kOwn) ? ws[4u * rank + 3u] != 0u : false;
  bool chg = false, found = false;
  uint32_t fbucket = rp_meta_bucket(emeta);
  uint32_t pusher = kNone;
  if (own) pusher = a.rec_pusher[code >> 5];
  uint32_t matched = 0;   // push j was matched by a record of mine
  for (uint32_t j = 0; j < n_lp; ++j) {
    const uint32_t lr = rp_hshfl(lp_rec, hb + j), llb = rp_hshfl(lp_lb, hb + j);
    if (own && pusher != kNone && lr == pusher && (llb & 0xFF) == rp_meta_lut(emeta)) {
      found = true; fbucket = llb >> 8; matched |= 1u << j;
    }
  }
  if (own) {
    const uint32_t r = code >> 5;
    uint32_t mn = emeta;
    if (pusher != kNone) {
      mn = rp_meta(rp_meta_lut(emeta), fbucket, found) | (emeta & (1u << 18));
      if (mn != emeta) chg = true;
    }
    if (rp_meta_live(emeta) && eT != kNever) {
      if (pop_moved) chg = true;
    } else {
      a.rec_d_n[r] = ed;
      a.rec_s_n[r] = es;
    }
    a.rec_meta_n[r] = mn;
    if (chg) rp_chg_push(a, r);
  }
  uint32_t any_matched = 0;
  for (uint32_t j = 0; j < n_lp; ++j)
    if ((uint32_t)(__ballot((matched >> j) & 1u) >> hb) & kGm) any_matched |= 1u << j;
  if (hl < n_lp && !((any_matched >> hl) & 1u)) {
    if (a.rec_kid[(size_t)lp_rec * 26 + (lp_lb & 0xFF)] == 0u) {
      uint32_t* bw = rp_born_slot(a, lp_rec);
      if (bw) {
        bw[0] = lp_rec;
        bw[1] = lp_lb & 0xFF;
        bw[2] = lp_lb >> 8;
        bw[3] = gid;
        bw[4] = __float_as_uint(lp_d);
        bw[5] = lp_s;
      }
    }
  }
}

// rp_fold_raise as one wave per target (same layout as rp_fold_wave)
__device__ inline void rp_fold_raise_wave(const rp::Args& a, uint32_t t, int lane, uint32_t* push_acc = nullptr) {
  using namespace rp;
  constexpr int Q = (int)kEvQ;
  Ctl& c = *a.ctl;
  const uint32_t gid = a.tgt_gid[t];
  if (gid == kNone) return;
  uint32_t n_all = a.tgt_cnt[t];
  if (n_all > a.c.ev) n_all = a.c.ev;
  const int slots = (int)((n_all + 63u) >> 6);
  uint32_t code[Q];
  unsigned long long eT[Q];
  bool valid[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    code[q] = 0; eT[q] = kNever; valid[q] = false;
    const uint32_t e = (uint32_t)lane + 64u * q;
    if (e < n_all) {
      code[q] = a.tgt_ev[(size_t)t * a.c.ev + e];
      eT[q] = a.rec_T[code[q] >> 5];
      valid[q] = (code[q] & 31) != kOwn && eT[q] < c.cut;
    }
  }
  uint32_t rank[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) rank[q] = 0;
  uint32_t n = 0;
#pragma unroll
  for (int q2 = 0; q2 < Q; ++q2) {
    if (q2 >= slots) break;
    unsigned long long vm = __ballot(valid[q2]);
    n += (uint32_t)__popcll(vm);
    while (vm) {
      const int k = __ffsll((long long)vm) - 1;
      vm &= vm - 1;
      const unsigned long long Tk = rl_u64(eT[q2], k);
#pragma unroll
      for (int q = 0; q < Q; ++q) rank[q] += (Tk < eT[q]) ? 1u : 0u;
    }
  }
  const float d0 = a.dist[gid];
  const uint32_t s0 = a.state[gid];
  float d = d0;
  uint32_t s = s0;
  const int RQ = a.c.num_buckets;
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t ecode = 0;
    bool got = false;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      if (q >= slots || got) continue;
      const unsigned long long m = __ballot(valid[q] && rank[q] == i);
      if (m) { ecode = rl_u32(code[q], __ffsll((long long)m) - 1); got = true; }
    }
    if (!got) break;
    const uint32_t r = ecode >> 5, lut = ecode & 31;
    bool to_raise;
    if (!rp_raise_event(a.c, &d, &s, (int)lut, &to_raise)) continue;
    if (lane == 0) {
      const int q = to_raise ? RQ : rp_bucket_of(a.c, d);
      const uint32_t w = r * 7 + lut / 4, sh = (lut % 4) * 8;
      atomicOr(&a.rec_push[w], (uint32_t)(q + 1) << sh);
      atomicAdd(push_acc ? &push_acc[q] : &c.push_cnt[q], 1u);
    }
  }
  if (lane == 0 && (d != d0 || s != s0)) {
    a.dist[gid] = d;
    a.state[gid] = s;
    rp::rp_mark_block(a, gid);
  }
}

// rp_fold_raise_wave for two targets at once, a half-wave each (lists of up to G events; see rp_fold_pair)
template <int G>
__device__ inline void rp_fold_raise_pair(const rp::Args& a, uint32_t t, uint32_t gid, uint32_t n_all, bool act, int lane, uint32_t* push_acc) {
  using namespace rp;
  Ctl& c = *a.ctl;
  constexpr uint32_t kGm = G == 32 ? 0xFFFFFFFFu : (1u << (G & 31)) - 1u;
  const uint32_t h = (uint32_t)lane / G, hl = (uint32_t)lane % G, hb = h * G;
  act = act && gid != kNone;
  if (!act) n_all = 0;
  uint32_t code = 0;
  unsigned long long eT = kNever;
  bool valid = false;
  if (hl < n_all) {
    code = a.tgt_ev[(size_t)t * a.c.ev + hl];
    eT = a.rec_T[code >> 5];
    valid = (code & 31) != kOwn && eT < c.cut;
  }
  const float d0 = act ? a.dist[gid] : 0.f;
  const uint32_t s0 = act ? a.state[gid] : 0u;
  const uint32_t key = (uint32_t)((eT >> kRankBits) << 11) | (uint32_t)(eT & 0x7FFull);
  const uint32_t vmask = (uint32_t)(__ballot(valid) >> hb) & kGm;
  const uint32_t n = (uint32_t)__popc(vmask);
  uint32_t rank = 0;
  for (uint32_t m = vmask; m; m &= m - 1u) {
    const uint32_t k = (uint32_t)__ffs((int)m) - 1u;
    rank += rp_hshfl(key, hb + k) < key ? 1u : 0u;
  }
  float d = d0;
  uint32_t s = s0;
  const int RQ = a.c.num_buckets;
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t m = (uint32_t)(__ballot(valid && rank == i) >> hb) & kGm;
    if (!m) break;
    const uint32_t ecode = rp_hshfl(code, hb + (uint32_t)__ffs((int)m) - 1u);
    const uint32_t r = ecode >> 5, lut = ecode & 31;
    bool to_raise;
    if (!rp_raise_event(a.c, &d, &s, (int)lut, &to_raise)) continue;
    if (hl == 0) {
      const int q = to_raise ? RQ : rp_bucket_of(a.c, d);
      const uint32_t w = r * 7 + lut / 4, sh = (lut % 4) * 8;
      atomicOr(&a.rec_push[w], (uint32_t)(q + 1) << sh);
      atomicAdd(push_acc ? &push_acc[q] : &c.push_cnt[q], 1u);
    }
  }
  if (hl == 0 && act && (d != d0 || s != s0)) {
    a.dist[gid] = d;
    a.state[gid] = s;
    rp::rp_mark_block(a, gid);
  }
}

// PH_SIM on the device: one workgroup per excursion.  The excursion's records (member list kept by PH_APPLY) are loaded
// into LDS in parallel — per record its pusher's position in the list, LUT index, bucket, liveness, its rank of the last
// ranking — and the queue discipline (rp_phase_sim, the serial form the CPU emulation runs) is replayed by wave 0 on LDS
// only: a pop costs a few hundred cycles instead of two dependent trips to L2.  The replay does not start from the base
// record: PH_APPLY keeps, per excursion, the smallest rank at which something changed (a child that appeared, died or
// moved to another bucket: Args::sub_restart); the pops up to that rank are as they were, the queue content at that point
// is rebuilt from the ranks (every live record whose pusher has popped and that has not popped itself, in arrival order =
// (pusher's rank, LUT index) per bucket) and only the pops behind it are replayed.  An excursion grows at its far end,
// so a ranking costs what changed, not what exists.
constexpr uint32_t kSimMax = 1024;   // records of one excursion the ranking handles (Cfg::smax <= this)
struct SimLds {
  // the live children of every record as CSR: kidw[first[pl] .. first[pl + 1]) in LUT order, pl = position of the pusher in
  // the member list + 1 (0: the base record)
  uint32_t cnt[kSimMax + 2];
  unsigned short first[kSimMax + 2];
  uint32_t kidw[kSimMax];                     // kids[]: position of the child in the member list | its bucket << 16
  uint32_t wave_tot[kRpThreads / 64];
  uint32_t info[kSimMax];                     // lut | bucket << 8 | live << 16 | poisoned << 17 | has an unlisted child << 18 | pusher's position << 19 (11 bits) | rp_moved_needs_mark << 30
  unsigned short next[kSimMax], rank[kSimMax];
  uint32_t pend_key[kSimMax];                 // pending records at the restart point: bucket << 24 | pusher rank << 5 | lut
  unsigned short pend_j[kSimMax], sorted[kSimMax];
  unsigned short head[rp::kMaxBuckets + 1], tail[rp::kMaxBuckets + 1];
  unsigned short moved[kSimMax];              // members whose pop time this ranking moved
  unsigned short orank[kSimMax];              // rank of the last ranking (0: none) | 0x8000: this ranking changed its order (rp_phase_sim, mark_moved = 2)
  uint32_t memr[kSimMax];                     // the members' record numbers (read once in pass 1: the write-back needs them again)
  uint32_t n_pend, flag_rank, n_ranked, truncated, n_moved;
};
}  // namespace
// The step kernel's LDS: a launch runs ONE phase, so the ranking's tables, the folds' per-wave event scratch, PH_APPLY's
// collector of dirty marks and the control block's copy (the control step runs when the phase is over) share one region —
// 34 KB, which with the counters below keeps four workgroups on a CU (the grid of 1,024 starts in one go: a launch with a
// second round of workgroups costs every phase twice its floor).  File scope, so that the core's hook reaches it.
struct RpWgDirty { uint32_t cnt, base; uint32_t list[kWgDirtyCap]; };
union RpLds {
  SimLds sim;
  uint32_t fold[kRpThreads / 64][kFoldLdsWords];
  RpWgDirty dirty;
  rp::Ctl ctl;
};
__shared__ RpLds g_rp_lds;
__shared__ uint32_t g_rp_pre[2 * (rp::kShards + 1)];   // PH_APPLY: Ctl::chg_pre, Ctl::born_pre
__shared__ const uint32_t* g_rp_pre_born;            // ... the address of Ctl::born_pre (tells the two arrays apart)
__shared__ uint32_t g_rp_collect;   // 1 while a phase runs whose marks are collected (outside the union: the hook reads it in every phase)
__device__ inline uint32_t rp_pre_lds(const uint32_t* arr, uint32_t k) {
  // (arr is Ctl::chg_pre or Ctl::born_pre of the block in memory: which one it is follows from its address)
  return g_rp_pre[(arr == g_rp_pre_born ? rp::kShards + 1 : 0) + k];
}
__device__ inline bool rp_wg_dirty_push(uint32_t t) {
  if (!g_rp_collect) return false;
  const uint32_t k = atomicAdd(&g_rp_lds.dirty.cnt, 1u);
  if (k >= kWgDirtyCap) return false;   // full: this mark goes to the shared list directly
  g_rp_lds.dirty.list[k] = t;
  return true;
}
namespace {

__device__ inline void rp_sim_block(const rp::Args& a, uint32_t base, SimLds& L) {
  rp::Ctl& c = *a.ctl;
  const uint32_t smax = a.c.smax;
  const int tid = threadIdx.x, lane = threadIdx.x & 63;
  const unsigned long long tk0 = wall_clock64();   // (100 MHz; Ctl::st_sim_ticks: tables built / queue replayed / pop times written)
  unsigned long long tk1 = tk0, tk2 = tk0;
  uint32_t n_batches = 0;
  // Every pass below issues ALL its loads before it uses one: a ranking is a chain of dependent trips to memory (~1.5 us each
  // on a mostly idle chip), and a loop of "load, test, load" over 256 members at a time pays the chain once per 256 members —
  // the launch waits for its LARGEST ranking (round 6: tables 7.8 + pop times 5.1 + marks 3.9 us for the average ranking of 25
  // members; a ranking that moved 300 members spent as long on its marks as on replaying the queue).
  constexpr int kPer1 = kSimMax / kRpThreads;
  // trip 1: the excursion's header and — when the list's place follows from the base record (Cfg::slot_by_base) — the member list
  const bool guess = a.c.slot_by_base && base < a.sub_slots_cap;
  uint32_t rr[kPer1];
  {
    const uint32_t* mem_g = a.sub_mem + (size_t)base * smax;
#pragma unroll
    for (int k = 0; k < kPer1; ++k) {
      const uint32_t j = tid + k * kRpThreads;
      rr[k] = (guess && j < smax) ? mem_g[j] : 0u;
    }
  }
  const uint32_t slot = a.sub_slot[base];
  const uint32_t mem_n = a.sub_mem_n[base];
  uint32_t p = a.sub_restart[base];          // ranks <= p stand
  const uint32_t old_n = a.sub_n[base];
  const uint32_t base_meta = a.rec_meta[base];
  const uint32_t it_now = (uint32_t)c.st_iters;
  uint32_t n = (slot != 0u && slot <= a.sub_slots_cap) ? mem_n : 0u;
  if (n > smax) n = smax;
  const int nb = (int)c.bucket;
  if (p > old_n) p = old_n;
  if (!(guess && slot == base + 1u)) {
    const uint32_t* mem = a.sub_mem + (size_t)(slot ? slot - 1 : 0) * smax;
#pragma unroll
    for (int k = 0; k < kPer1; ++k) {
      const uint32_t j = tid + k * kRpThreads;
      rr[k] = j < n ? mem[j] : 0u;
    }
  }
  __syncthreads();
  const unsigned long long ta = wall_clock64();
  for (uint32_t i = tid; i < n + 2; i += kRpThreads) L.cnt[i] = 0;
  for (int i = tid; i < nb; i += kRpThreads) { L.head[i] = 0xFFFF; L.tail[i] = 0xFFFF; }
  if (tid == 0) { L.n_pend = 0; L.flag_rank = (base_meta & (1u << 18)) ? 0u : 0xFFFFFFFFu; }
  // trip 2: the members' records (a member always has a pusher — PH_APPLY made it —, so rp::rp_moved_needs_mark is its birth
  // iteration alone: bit 30 of the table word, and the write-back does not go to memory for it)
  uint32_t m4[kPer1], pl4[kPer1], po4[kPer1], bi4[kPer1];
  unsigned long long T4[kPer1];
#pragma unroll
  for (int k = 0; k < kPer1; ++k) {
    const uint32_t j = tid + k * kRpThreads;
    const bool on = j < n;
    const uint32_t r = rr[k];
    m4[k] = on ? a.rec_meta[r] : 0u;
    pl4[k] = on ? (a.rec_plocal ? a.rec_plocal[r] : a.rec_local[a.rec_pusher[r]]) : 0u;
    T4[k] = on ? a.rec_T[r] : rp::kNever;
    po4[k] = on ? a.rec_poison[r] : 0u;
    bi4[k] = (on && a.rec_born_it) ? a.rec_born_it[r] : ~it_now;
  }
  __syncthreads();
  const unsigned long long tb = wall_clock64();
  // pass 1: records, their old ranks, the child table
#pragma unroll
  for (int k = 0; k < kPer1; ++k) {
    const uint32_t j = tid + k * kRpThreads;
    if (j >= n) continue;
    const uint32_t m = m4[k], pl = pl4[k];
    const unsigned long long T = T4[k];
    L.memr[j] = rr[k];
    uint32_t rk = 0xFFFF;
    if (T != rp::kNever && (uint32_t)(T & rp::kRankMask) <= p) rk = (uint32_t)(T & rp::kRankMask);   // it popped in front of the restart point
    L.info[j] = (m & 0x1FFFFu) | (po4[k] ? (1u << 17) : 0u) | (m & (1u << 18)) | (pl << 19) | (bi4[k] == it_now ? 0u : (1u << 30));
    L.next[j] = (unsigned short)0xFFFF;
    L.rank[j] = (unsigned short)rk;
    L.orank[j] = T != rp::kNever ? (unsigned short)(T & 0x7FFFu) : (unsigned short)0;   // (ranks stay below smax <= 1024)
    if (rp::rp_meta_live(m)) atomicOr(&L.cnt[pl], 1u << (m & 0x1Fu));   // the LUT indices of pl's live children (a pusher has one child per neighbour)
    if (rk != 0xFFFF && (m & (1u << 18))) atomicMin(&L.flag_rank, rk);   // ranked, but a child of it is not in the list
  }
  __syncthreads();
  {
    // first[] = exclusive prefix of the children per record (bits of cnt[0 .. n]): five consecutive entries per thread, wave scan, wave totals
    constexpr int kPer = (kSimMax + 2 + kRpThreads - 1) / kRpThreads;
    uint32_t v[kPer], sum = 0;
    for (int k = 0; k < kPer; ++k) {
      const uint32_t i = tid * kPer + k;
      v[k] = i <= n ? (uint32_t)__popc(L.cnt[i]) : 0u;
      sum += v[k];
    }
    uint32_t inc = sum;
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(inc, d);
      if (lane >= d) inc += o;
    }
    if (lane == 63) L.wave_tot[tid >> 6] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < (tid >> 6); ++w) before += L.wave_tot[w];
    uint32_t run = before + inc - sum;
    for (int k = 0; k < kPer; ++k) {
      const uint32_t i = tid * kPer + k;
      if (i <= n + 1) L.first[i] = (unsigned short)run;
      run += v[k];
    }
  }
  __syncthreads();
  // a record's children in LUT order: a child's place is the number of its siblings with a smaller LUT index.  (Until round 6 the
  // children were filed in arrival order and every record's run was insertion-sorted through LDS — two dependent reads per
  // comparison, 26 children of a base record ~170 of them; the rankings a launch waits for spent as long here as in the replay.)
  for (uint32_t j = tid; j < n; j += kRpThreads) {
    const uint32_t inf = L.info[j];
    const uint32_t pl = (inf >> 19) & 0x7FFu;
    if ((inf >> 16) & 1u) L.kidw[L.first[pl] + (uint32_t)__popc(L.cnt[pl] & ((1u << (inf & 0x1Fu)) - 1u))] = j | (((inf >> 8) & 0xFFu) << 16);
  }
  __syncthreads();
  const bool arrays = nb <= 64;   // the bucket FIFOs as arrays (below); more buckets than lanes: linked lists, one pop at a time
  if (arrays && tid < 64) { L.cnt[tid] = 0; L.tail[tid] = 0; L.moved[tid] = 0; }   // (cnt is free since the child table was filled)
  // a record with an unlisted child in front of the restart point: the ranking ends behind it
  uint32_t R = p;
  bool truncated = false;
  if (L.flag_rank <= p) { R = L.flag_rank; truncated = true; }
  __syncthreads();
  const unsigned long long te = wall_clock64();
  // pass 2: the queue at the restart point
  if (!truncated) {
    for (uint32_t j = tid; j < n; j += kRpThreads) {
      const uint32_t inf = L.info[j];
      if (!((inf >> 16) & 1u)) continue;                      // dead
      if (L.rank[j] != 0xFFFF) continue;                      // popped already
      if (arrays && ((inf >> 8) & 0xFF) < 64u) atomicAdd(&L.cnt[(inf >> 8) & 0xFF], 1u);   // it may enter its bucket's FIFO in this replay: room for it
      const uint32_t pl = (inf >> 19) & 0x7FFu;
      const uint32_t pr = pl == 0 ? 0u : (uint32_t)L.rank[pl - 1];
      if (pl != 0 && (pr == 0xFFFF)) continue;                // its pusher has not popped
      const uint32_t k = atomicAdd(&L.n_pend, 1u);
      L.pend_key[k] = (((inf >> 8) & 0xFF) << 24) | (pr << 5) | (inf & 0x1F);
      L.pend_j[k] = (unsigned short)j;
    }
  } else {
    for (uint32_t j = tid; j < n; j += kRpThreads)
      if (L.rank[j] != 0xFFFF && L.rank[j] > R) L.rank[j] = 0xFFFF;
  }
  __syncthreads();
  const uint32_t P = L.n_pend;
  for (uint32_t k = tid; k < P; k += kRpThreads) {
    const uint32_t key = L.pend_key[k];
    uint32_t pos = 0;
    for (uint32_t q = 0; q < P; ++q) pos += (L.pend_key[q] < key) ? 1u : 0u;   // keys are distinct: (pusher, lut) is
    L.sorted[pos] = L.pend_j[k];
  }
  __syncthreads();
  if (arrays) {
    // Bucket q's FIFO is the array L.next[L.head[q] ..): room for every live record of that bucket that has not popped (a record
    // enters a FIFO once).  The pending records are sorted by (bucket, pusher's rank, LUT index): run q of `sorted` is the FIFO's
    // content at the restart point — L.tail[q] / L.moved[q]: where the run starts / ends.
    if (tid < 64) {
      const uint32_t cq = tid < nb ? L.cnt[tid] : 0u;
      L.head[tid] = (unsigned short)(rp_wave_scan_add(cq) - cq);
    }
    for (uint32_t k = tid; k < P; k += kRpThreads) {
      const uint32_t kb = (L.info[L.sorted[k]] >> 8) & 0xFF;
      if (k == 0 || ((L.info[L.sorted[k - 1]] >> 8) & 0xFF) != kb) L.tail[kb] = (unsigned short)k;
      if (k + 1 == P || ((L.info[L.sorted[k + 1]] >> 8) & 0xFF) != kb) L.moved[kb] = (unsigned short)(k + 1);
    }
    __syncthreads();
    for (uint32_t k = tid; k < P; k += kRpThreads) {
      const uint32_t j = L.sorted[k];
      const uint32_t kb = (L.info[j] >> 8) & 0xFF;
      L.next[L.head[kb] + k - L.tail[kb]] = (unsigned short)j;
    }
  } else {
    for (uint32_t k = tid; k < P; k += kRpThreads) {
      const uint32_t j = L.sorted[k];
      const uint32_t kb = (L.info[j] >> 8) & 0xFF;
      const bool first = k == 0 || ((L.info[L.sorted[k - 1]] >> 8) & 0xFF) != kb;
      const bool last = k + 1 == P || ((L.info[L.sorted[k + 1]] >> 8) & 0xFF) != kb;
      L.next[j] = last ? (unsigned short)0xFFFF : L.sorted[k + 1];
      if (first) L.head[kb] = (unsigned short)j;
      if (last) L.tail[kb] = (unsigned short)j;
    }
  }
  __syncthreads();
  tk1 = wall_clock64();
  if (tid < 64 && !truncated && arrays) {
    // Replay of the queue discipline from the restart point by ONE wave, A BATCH OF POPS AT A TIME.  Everything that sits in the
    // lowest non-empty bucket pops before anything else does — children enter at the tail of their bucket, and a child's bucket
    // is below its pusher's only when the pusher's distance fell after it was queued — so up to 64 entries of that FIFO pop
    // together: lane k takes entry k, ranks are rank + k + 1, "was overtaken" is a prefix maximum over the old ranks, and the
    // children of all of them are appended in (pusher, LUT) order by counting, per bucket, the children in front of each.  The
    // batch is cut where the serial loop would behave differently: in front of a poisoned record / the rank limit (the ranking
    // ends), behind a record with an unlisted child (it ends), behind a record with a child in a LOWER bucket (that child is
    // next).  Until round 6 this was one pop per trip through the loop: 0.32 us per pop — 60 instructions of one wave, a dozen
    // taken branches, two LDS waits —, and a launch waits for its largest ranking (first update of a map: 2,500 rankings of more
    // than 256 pops); the stream's rankings pop 5.9 entries per batch on average.
    // Lane q holds bucket q's FIFO: its place in L.next (qs), entries popped (hd) and entered (tl).
    uint32_t rank = R;
    uint32_t runmax = R;   // largest old rank among the records that have popped (the records up to the restart point kept theirs: 1 .. R)
    const uint32_t qs = lane < nb ? (uint32_t)L.head[lane] : 0u;
    uint32_t hd = 0, tl = lane < nb ? (uint32_t)L.moved[lane] - (uint32_t)L.tail[lane] : 0u;
    uint32_t* scr = L.pend_key;   // (the pending records' keys are dead) children of the batch: record | bucket << 16
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (;;) {
      const unsigned long long nonempty = __ballot(hd < tl);
      if (!nonempty) break;                                      // BucketQueue::front / pop
      const int b = __ffsll((long long)nonempty) - 1;
      const uint32_t at = rl_u32(qs + hd, b), m0 = rl_u32(tl - hd, b);
      const uint32_t m = m0 < 64u ? m0 : 64u;
      ++n_batches;
      const bool in = (uint32_t)lane < m;
      const uint32_t j = in ? (uint32_t)L.next[at + lane] : 0u;
      uint32_t inf = 0, o = 0, f = 0, e = 0;
      if (in) { inf = L.info[j]; o = (uint32_t)L.orank[j] & 0x7FFFu; f = L.first[j + 1]; e = L.first[j + 2]; }
      const uint32_t nk = e - f;
      const unsigned long long mA = __ballot(in && ((inf & (1u << 17)) || rank + (uint32_t)lane >= smax - 1));   // the ranking ends in front of it
      const unsigned long long mB = __ballot(in && (inf & (1u << 18)));                                          // ... behind it
      const uint32_t cutA = mA ? (uint32_t)__ffsll((long long)mA) - 1u : 65u, cutB = mB ? (uint32_t)__ffsll((long long)mB) : 65u;   // (65: none)
      const uint32_t inc = rp_wave_scan_add(nk);
      const unsigned long long mD = __ballot(in && inc > kSimMax);   // more children than the scratch holds: the batch ends in front of it
      const uint32_t cutD = mD ? (uint32_t)__ffsll((long long)mD) - 1u : 65u;
      uint32_t M = m < cutA ? m : cutA;
      M = M < cutB ? M : cutB;
      M = M < cutD ? M : cutD;
      // the children of the batch, in (pusher, LUT) order
      bool low = false;
      const uint32_t off = inc - nk;
      for (uint32_t c0 = 0;; c0 += 4) {   // (four at a time: the reads of a trip go out together)
        if (!__ballot((uint32_t)lane < M && c0 < nk)) break;
        uint32_t w4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w4[k] = ((uint32_t)lane < M && c0 + k < nk) ? L.kidw[f + c0 + k] : 0xFFFF0000u;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if ((uint32_t)lane < M && c0 + k < nk) {
            scr[off + c0 + k] = w4[k];
            low = low || (w4[k] >> 16) < (uint32_t)b;
          }
      }
      const unsigned long long mC = __ballot(low);   // a child in a bucket below this one pops next
      const uint32_t cutC = mC ? (uint32_t)__ffsll((long long)mC) : 65u;
      M = M < cutC ? M : cutC;
      // ranks; it pops now and did not before, or a record that used to pop behind it has popped in front of it: its order moved
      const bool pops = (uint32_t)lane < M;
      const uint32_t omax = rp_wave_scan_max(pops ? o : 0u);
      uint32_t before = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)omax, 0x138, 0xf, 0xf, false);   // wave_shr:1 (lane 0: 0)
      before = before > runmax ? before : runmax;
      if (pops) {
        L.rank[j] = (unsigned short)(rank + lane + 1);
        if (o == 0u || o < before) L.orank[j] = (unsigned short)(o | 0x8000u);
      }
      {
        const uint32_t all = rl_u32(omax, 63);
        if (all > runmax) runmax = all;
      }
      rank += M;
      if (cutB == M) { truncated = true; break; }                  // a child of the last one is not in the list: stop behind it
      if (cutC != M && cutA == M) { truncated = true; break; }     // the next one is poisoned / beyond the rank limit
      // the children of the M records enter their buckets: per bucket present, the lanes count who is in front of them
      const uint32_t T = M ? rl_u32(inc, (int)M - 1) : 0u;
      for (uint32_t x0 = 0; x0 < T; x0 += 64) {
        const bool has = x0 + lane < T;
        const uint32_t v = has ? scr[x0 + lane] : 0xFFFFFFFFu;
        const uint32_t kb = v >> 16;
        unsigned long long rem = __ballot(has);
        while (rem) {
          const uint32_t q = rl_u32(kb, __ffsll((long long)rem) - 1);
          const unsigned long long same = __ballot(has && kb == q);
          const uint32_t to = rl_u32(qs + tl, (int)q);
          if (has && kb == q) L.next[to + __popcll(same & lt)] = (unsigned short)(v & 0xFFFFu);
          if ((uint32_t)lane == q) tl += (uint32_t)__popcll(same);
          rem &= ~same;
        }
      }
      if (lane == b) hd += M;
    }
    R = rank;
  } else if (tid < 64 && !truncated) {
    // (more than 64 buckets: heads and tails stay in LDS)
    uint32_t rank = R;
    uint32_t runmax = R;
    int lowest = 0;
    for (;;) {
      while (lowest < nb && L.head[lowest] == 0xFFFF) ++lowest;   // BucketQueue::front / pop
      if (lowest >= nb) break;
      const uint32_t j = L.head[lowest];
      const uint32_t nx = L.next[j];
      const uint32_t inf = L.info[j];
      rp_wave_sync();
      L.head[lowest] = (unsigned short)nx;
      if (nx == 0xFFFF) L.tail[lowest] = 0xFFFF;
      if (rank >= smax - 1 || (inf & (1u << 17))) { truncated = true; break; }
      ++rank;
      L.rank[j] = (unsigned short)rank;
      {
        const uint32_t o = L.orank[j] & 0x7FFFu;
        if (o == 0u || o < runmax) L.orank[j] = (unsigned short)(o | 0x8000u);
        if (o > runmax) runmax = o;
      }
      rp_wave_sync();
      if (inf & (1u << 18)) { truncated = true; break; }
      uint32_t jv = 0xFFFF, jinfo = 0;
      {
        const uint32_t f = L.first[j + 1], e = L.first[j + 2];
        if ((uint32_t)lane < e - f) {
          jv = L.kidw[f + lane] & 0xFFFFu;
          jinfo = L.info[jv];
        }
      }
      unsigned long long kids = __ballot(jv != 0xFFFF);
      while (kids) {
        const int src = __ffsll((long long)kids) - 1;
        kids &= kids - 1;
        const uint32_t cj = __shfl(jv, src);
        const int kb = (int)((__shfl(jinfo, src) >> 8) & 0xFF);
        const uint32_t tl = L.tail[kb];
        L.next[cj] = 0xFFFF;
        if (tl == 0xFFFF) L.head[kb] = (unsigned short)cj; else L.next[tl] = (unsigned short)cj;
        L.tail[kb] = (unsigned short)cj;
        if (kb < lowest) lowest = kb;
        rp_wave_sync();
      }
    }
    R = rank;
  }
  tk2 = wall_clock64();
  uint32_t popped = 0;
  if (tid == 0) {
    popped = R - (L.flag_rank <= p ? L.flag_rank : p);   // pops this ranking replayed (behind the restart point)
    L.n_ranked = R;
    L.truncated = truncated ? 1u : 0u;
  }
  __syncthreads();
  if (tid == 0) L.n_moved = 0;
  __syncthreads();
  for (uint32_t j = tid; j < n; j += kRpThreads) {
    const uint32_t rk = L.rank[j];
    const uint32_t r = L.memr[j];
    const unsigned long long Tn = rk == 0xFFFF ? rp::kNever : (((unsigned long long)base << rp::kRankBits) | rk);
    // a pop time that moved reorders the events of the targets the record talks to (rp::rp_mark_rec_targets).  The old pop
    // time is (base, old rank) — a record never leaves its excursion —, and the old rank is in the table since pass 1 (0: it
    // had not popped): no second trip to rec_T
    const uint32_t o_rk = L.orank[j] & 0x7FFFu, n_rk = rk == 0xFFFF ? 0u : rk;
    bool moved = o_rk != n_rk;                                                  // mark_moved = 1: every pop time that moved
    if (a.c.mark_moved >= 2) moved = rk == 0xFFFF ? (o_rk != 0u) : ((L.orank[j] & 0x8000u) != 0u);   // 2: order changes only (rp_phase_sim)
    if (a.c.mark_moved && moved && (L.info[j] & (1u << 30))) L.moved[atomicAdd(&L.n_moved, 1u)] = (unsigned short)j;   // (bit 30: rp::rp_moved_needs_mark)
    if (o_rk != n_rk) a.rec_T[r] = Tn;
  }
  __syncthreads();
  const unsigned long long tk3 = wall_clock64();
  // one (moved record, target) pair per thread: 27 dependent atomics in a row on one lane would be most of a small ranking's time.
  // The marks are collected in LDS (the pending-queue tables are dead by now) and filed with ONE atomic on Ctl::n_dirty per
  // ranking: every wave with a mark used to increment that word — eight increments per ranking, a hundred rankings per launch,
  // on an address that takes ~90 atomics per microsecond: two thirds of a ranking's time (st_sim_ticks).
  if (tid == 0) L.n_pend = 0;
  __syncthreads();
  const uint32_t wl = 1u - c.read;
  {
    // kMarkB (moved record, target) pairs per thread at a time: their targets in one trip, their marks in a second
    constexpr int kMarkB = 8;
    const uint32_t pairs = L.n_moved * 27u;
    for (uint32_t i0 = 0; i0 < pairs; i0 += kMarkB * kRpThreads) {
      uint32_t t8[kMarkB], was[kMarkB];
#pragma unroll
      for (int k = 0; k < kMarkB; ++k) {
        const uint32_t i = i0 + k * kRpThreads + tid;
        t8[k] = i < pairs ? a.rec_tgts[(size_t)L.memr[L.moved[i / 27u]] * 27 + i % 27u] : rp::kSkip;
      }
#pragma unroll
      for (int k = 0; k < kMarkB; ++k) was[k] = t8[k] < rp::kSkip ? atomicExch(&a.tgt_dirty[t8[k]], 1u) : 1u;
#pragma unroll
      for (int k = 0; k < kMarkB; ++k) {
        if (was[k] != 0u) continue;
        const uint32_t q = atomicAdd(&L.n_pend, 1u);
        if (q < kSimMax) L.pend_key[q] = t8[k];
        else a.dl[wl][atomicAdd(&c.n_dirty[wl], 1u)] = t8[k];
      }
    }
  }
  __syncthreads();
  const unsigned long long tk4 = wall_clock64();
  {
    const uint32_t nd = L.n_pend < kSimMax ? L.n_pend : kSimMax;
    if (tid == 0 && nd) L.flag_rank = atomicAdd(&c.n_dirty[wl], nd);
    __syncthreads();
    for (uint32_t i = tid; i < nd; i += kRpThreads) a.dl[wl][L.flag_rank + i] = L.pend_key[i];
  }
  if (tid == 0) {
    a.sub_dirty[base] = 0;
    a.sub_restart[base] = rp::kNone;
    a.sub_n[base] = L.n_ranked;
    if (L.truncated) {
      atomicMin(&c.smax_cut, ((unsigned long long)base << rp::kRankBits) | (L.n_ranked + 1));
      atomicAdd(&c.st_trunc_rank, 1ull);
    }
    if (a.wg_stats && blockIdx.x < 4096u) {   // (kept per workgroup: atomics on Ctl's lines slowed down what they measured)
      const unsigned long long tk5 = wall_clock64();
      unsigned long long* w = a.wg_stats + (size_t)blockIdx.x * rp::kWgStats;
      w[0] += n;
      w[1] += popped;
      w[2 + (popped < 16u ? 0 : popped < 64u ? 1 : popped < 256u ? 2 : 3)] += 1;
      w[6] += ta - tk0;
      w[7] += tb - ta;
      w[8] += te - tb;
      w[9] += tk1 - te;
      w[10] += tk2 - tk1;
      w[11] += tk3 - tk2;
      w[12] += tk4 - tk3;
      w[13] += tk5 - tk4;
      w[14] += n_batches;
      w[15] += 1;
      {   // rankings by what they took: < 8, < 12, < 16, < 24, < 32 us, more; and what the ones of 16 us or more were made of
        const unsigned long long all = tk5 - tk0;   // (10 ns units)
        w[16 + (all < 800 ? 0 : all < 1200 ? 1 : all < 1600 ? 2 : all < 2400 ? 3 : all < 3200 ? 4 : 5)] += 1;
        if (all >= 1600) {
          w[22] += tb - tk0; w[23] += tk1 - tb; w[24] += tk2 - tk1; w[25] += tk5 - tk2;
          w[26] += n; w[27] += n_batches; w[28] += popped;
        }
      }
    }
  }
  __syncthreads();
}

// SCAN: tiles of kRpThreads items; exclusive prefix of the first nc (<= rp::kScanC) counts per item, F::count / F::apply per item (with
// an F::State between them: what count() found out need not be asked again), totals -> tot[0..nc) (atomic stores), a wait that
// does not end -> *err |= 64.  Tiles are handed out by ticket, or — by_block, for a grid that is resident as a whole, tile t's
// predecessors are then running or done as well — tile = workgroup: two trips to a shared counter less per workgroup.
// Round 6, after the stages were timed per tile (VBX_RP_STATS: ticket 0.8, counts 9, scan 5, look-back 5-10, apply 9.7, last
// ticket 2 us): the scan inside the tile on the DPP path with the per-component loops unrolled (runtime-indexed register arrays
// had gone to scratch), a wave's two components looked back together, the reduction of a look-back window on the DPP path.
template <class F>
__device__ inline void rp_scan_tiles(const F& f, const RpScan& sc, uint32_t n, uint32_t* tot, uint32_t* err, int nc = rp::kScanC, bool by_block = false,
                                     unsigned long long* stats = nullptr) {
  __shared__ uint32_t s_tile;
  __shared__ uint32_t s_wave[kRpThreads / 64][rp::kScanC];
  __shared__ uint32_t s_prefix[rp::kScanC];
  constexpr int kWaves = kRpThreads / 64;
  constexpr int kPerWave = (rp::kScanC + kWaves - 1) / kWaves;   // components a wave publishes and looks back for
  const uint32_t tiles = (n + kRpThreads - 1) / kRpThreads;
  if (tiles > sc.max_tiles) {   // more tiles than descriptors: fail loudly instead of indexing past the buffer
    if (threadIdx.x == 0) atomicOr(err, 64u);
    return;
  }
  const uint32_t gen = sc.ticket[1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t round = 0;; ++round) {
    const unsigned long long sk0 = wall_clock64();
    uint32_t tile;
    if (by_block) {
      tile = blockIdx.x + round * gridDim.x;
      if (tile >= tiles) break;
      if (round) __syncthreads();   // (the tables below are used again)
    } else {
      __syncthreads();
      if (threadIdx.x == 0) s_tile = atomicAdd(&sc.ticket[0], 1u);
      __syncthreads();
      tile = s_tile;
      if (tile >= tiles) { if (stats && threadIdx.x == 0) stats[5] += wall_clock64() - sk0; break; }
    }
    const unsigned long long sk1 = wall_clock64();
    const uint32_t i = tile * kRpThreads + threadIdx.x;
    rp::Cnt4 cnt{};
    typename F::State st{};
    if (i < n) cnt = f.count(i, st);
    const unsigned long long sk2 = wall_clock64();
    // exclusive scan inside the tile: wave scans on the DPP path, wave totals through LDS
    rp::Cnt4 ex{};
#pragma unroll
    for (int k = 0; k < (int)rp::kScanC; ++k) {
      if (k < nc) {
        const uint32_t inc = rp_wave_scan_add(cnt.v[k]);
        if (lane == 63) s_wave[wave][k] = inc;
        ex.v[k] = inc - cnt.v[k];
      }
    }
    __syncthreads();
    // the components this wave publishes and looks back for: wave, wave + kWaves, ...
    uint32_t agg[kPerWave];
#pragma unroll
    for (int j = 0; j < kPerWave; ++j) agg[j] = 0;
#pragma unroll
    for (int k = 0; k < (int)rp::kScanC; ++k) {
      if (k < nc) {
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
          const uint32_t x = s_wave[w][k];
          if (w < wave) before += x;
          total += x;
        }
        ex.v[k] += before;
        if (k % kWaves == wave) agg[k / kWaves] = total;   // (k / kWaves is a constant of the unrolled loop)
      }
    }
    const unsigned long long sk3 = wall_clock64();
    // publish the aggregates, look back 64 predecessors at a time (a tile's wait is for aggregates only, which every tile
    // publishes before it looks back — no chain of waits through the tiles); the wave's components look back together
    const unsigned long long tag_agg = (unsigned long long)((gen << 2) | (tile == 0 ? 2u : 1u)) << 32, tag_inc = (unsigned long long)((gen << 2) | 2u) << 32;
    uint32_t prefix[kPerWave], left[kPerWave];   // tiles [0, left) are still to be summed
#pragma unroll
    for (int j = 0; j < kPerWave; ++j) {
      const int k = wave + j * kWaves;
      prefix[j] = 0;
      left[j] = k < nc ? tile : 0u;
      if (lane == 0 && k < nc) __hip_atomic_store(sc.desc + (size_t)tile * rp::kScanC + k, tag_agg | agg[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    uint32_t spins = 0;
    // one window of one component: true when it was summed (t moves on, or to 0 behind an inclusive prefix)
    const auto window = [&](unsigned long long w, uint32_t& t, uint32_t& pre) -> bool {
      const bool mine = (uint32_t)lane < t;
      const uint32_t tag = (uint32_t)(w >> 32);
      const bool ready = mine && (tag >> 2) == gen && (tag & 3u) != 0u;
      const bool incl = ready && (tag & 3u) == 2u;
      const unsigned long long m_incl = __ballot(incl), m_ready = __ballot(ready), m_mine = __ballot(mine);
      // lanes up to the first inclusive prefix (or all of mine) must be there
      const int stop = m_incl ? (__ffsll((long long)m_incl) - 1) : 63;
      const unsigned long long need = (stop == 63 ? ~0ull : ((2ull << stop) - 1ull)) & m_mine;
      if ((m_ready & need) != need) return false;
      pre += rl_u32(rp_wave_scan_add(((need >> lane) & 1ull) ? (uint32_t)w : 0u), 63);
      t = m_incl ? 0u : t - (uint32_t)__popcll(m_mine);
      return true;
    };
    for (;;) {
      bool any = false;
#pragma unroll
      for (int j = 0; j < kPerWave; ++j) any = any || left[j] > 0;
      if (!any) break;
      unsigned long long w[kPerWave];
#pragma unroll
      for (int j = 0; j < kPerWave; ++j) {
        w[j] = 0;
        if ((uint32_t)lane < left[j])
          w[j] = __hip_atomic_load(sc.desc + (size_t)(left[j] - 1 - lane) * rp::kScanC + (wave + j * kWaves), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      bool moved = false;
#pragma unroll
      for (int j = 0; j < kPerWave; ++j)
        if (left[j] > 0) moved = window(w[j], left[j], prefix[j]) || moved;
      if (!moved) {
        if (++spins > kRpSpinMax) { if (lane == 0) atomicOr(err, 64u); break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < kPerWave; ++j) {
        const int k = wave + j * kWaves;
        if (k < nc) {
          if (tile != 0) __hip_atomic_store(sc.desc + (size_t)tile * rp::kScanC + k, tag_inc | (prefix[j] + agg[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s_prefix[k] = prefix[j];
          if (tile == tiles - 1) atomicExch(&tot[k], prefix[j] + agg[j]);
        }
      }
    }
    __syncthreads();
    const unsigned long long sk4 = wall_clock64();
    if (i < n) {
#pragma unroll
      for (int k = 0; k < (int)rp::kScanC; ++k)
        if (k < nc) ex.v[k] += s_prefix[k];
      f.apply(i, ex, st);
    }
    if (stats && threadIdx.x == 0) {   // (VBX_RP_STATS) ticket, counts, scan inside the tile, look-back, apply (issue only), the ticket that ends the loop
      stats[0] += sk1 - sk0; stats[1] += sk2 - sk1; stats[2] += sk3 - sk2; stats[3] += sk4 - sk3; stats[4] += wall_clock64() - sk4; stats[6] += 1; stats[7] += (unsigned long long)nc;
    }
  }
  if (n == 0 && blockIdx.x == 0 && (int)threadIdx.x < nc) atomicExch(&tot[threadIdx.x], 0u);
}

// PH_RANK / PH_PUSH.  The push pass on the device does not walk rp::rp_for_pushes twice: count() fetches the record's seven push
// words at once, finds the (few: 1.3 on average) pushes of this pass with the buckets' components from an LDS table, asks for
// their children's records four at a time, and leaves what it found — LUT indices taken, their components — to apply().  (The
// serial form went to memory once per word, then per push for the child, then for the child's record: 9 us per tile and stage.)
struct RpPhaseScan {
  const rp::Args& a;
  uint32_t phase;
  const uint32_t* comp_of;     // LDS [256]: bucket -> component of this pass (0xFF: another pass)
  const uint32_t* comp_bucket; // LDS [kScanC]: the component's bucket,
  const uint32_t* comp_tail;   // ... that bucket's FIFO tail
  bool push_last;              // the pass cleans up behind itself: a record's push words, children and poison flag are cleared by the thread that has just read them
  struct State { uint32_t take, kk[4], r, gid; };   // kk: the component of LUT index l in bits 4 (l % 8) .. of word l / 8
  static_assert(rp::kScanC <= 16, "four bits per component");
  __device__ static uint32_t kk_get(const State& st, uint32_t lut) { return (st.kk[lut / 8] >> (4 * (lut % 8))) & 15u; }
  __device__ rp::Cnt4 count(uint32_t i, State& st) const {
    const rp::Ctl& c = *a.ctl;
    rp::Cnt4 n{};
    if (phase == rp::PH_RANK) {
      const unsigned long long T0 = (unsigned long long)i << rp::kRankBits;
      if (T0 < c.cut) {
        n.v[0] = 1 + a.sub_n[i];
        if ((c.cut >> rp::kRankBits) == i) n.v[0] = (uint32_t)(c.cut & rp::kRankMask);  // ranks in front of the cut's, plus the base record
      }
      return n;
    }
    const uint32_t r = a.ord[i];
    st.r = r;
    st.gid = a.rec_vox[r];
    uint32_t w[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) w[k] = a.rec_push[r * 7 + k];
    // candidates: the pushes into a bucket of this pass
    uint32_t cand = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      uint32_t word = w[k];
      for (uint32_t j = 0; word != 0u; ++j, word >>= 8) {
        const uint32_t v = word & 0xFFu;
        if (v == 0u) continue;
        const uint32_t comp = comp_of[v - 1u];
        if (comp == 0xFFu) continue;
        const uint32_t lut = (uint32_t)k * 4 + j;
        cand |= 1u << lut;
        st.kk[lut / 8] |= comp << (4 * (lut % 8));
      }
    }
    // an entry that was popped inside this super-step is not queued: four candidates' children per trip
    const unsigned long long cut = c.cut;
    while (cand) {
      uint32_t lut4[4], kid4[4], m4[4];
      unsigned long long T4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        lut4[q] = cand ? (uint32_t)__ffs((int)cand) - 1u : 32u;
        if (cand) cand &= cand - 1u;
        kid4[q] = lut4[q] < 32u ? a.rec_kid[(size_t)r * 26 + lut4[q]] : 0u;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        m4[q] = kid4[q] ? a.rec_meta[kid4[q] - 1u] : 0u;
        T4[q] = kid4[q] ? a.rec_T[kid4[q] - 1u] : rp::kNever;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (lut4[q] >= 32u) continue;
        if (kid4[q] != 0u && rp::rp_meta_live(m4[q]) && T4[q] < cut) continue;
        st.take |= 1u << lut4[q];
        const uint32_t comp = kk_get(st, lut4[q]);
#pragma unroll
        for (int k = 0; k < (int)rp::kScanC; ++k)
          if ((uint32_t)k == comp) ++n.v[k];
      }
    }
    return n;
  }
  __device__ void apply(uint32_t i, const rp::Cnt4& prefix, const State& st) const {
    if (phase == rp::PH_RANK) {
      a.off0[i] = prefix.v[0];
      return;
    }
    rp::Cnt4 pos = prefix;   // (LUT indices come in ascending order: the entries of a bucket land in LUT order)
    uint32_t take = st.take;
    while (take) {
      const uint32_t lut = (uint32_t)__ffs((int)take) - 1u;
      take &= take - 1u;
      const uint32_t comp = kk_get(st, lut);
      uint32_t at = 0;
#pragma unroll
      for (int k = 0; k < (int)rp::kScanC; ++k)
        if ((uint32_t)k == comp) at = pos.v[k]++;
      rp::rp_queue_store(a, (int)comp_bucket[comp], comp_tail[comp] + at, rp::rp_neighbour(a, st.gid, (int)lut));
    }
    if (push_last) {   // (rp::rp_phase_cleanup's part for this record: nobody else reads its words in this launch)
      const uint32_t r = st.r;
#pragma unroll
      for (int k = 0; k < 26; ++k) a.rec_kid[(size_t)r * 26 + k] = 0;
#pragma unroll
      for (int k = 0; k < 7; ++k) a.rec_push[r * 7 + k] = 0;
      a.rec_poison[r] = 0;
    }
  }
};
__device__ inline void rp_scan_phase(const rp::Args& a, const RpScan& sc, uint32_t n, uint32_t phase) {
  __shared__ uint32_t s_comp_of[256];
  __shared__ uint32_t s_comp_bucket[rp::kScanC], s_comp_tail[rp::kScanC];
  // PH_RANK counts one thing per item, a PH_PUSH pass one per bucket it fills (Ctl::push_n: part A, written by the control
  // step of an earlier launch)
  const rp::Ctl& c = *a.ctl;
  int nc = phase == rp::PH_RANK ? 1 : (int)c.push_n;
  if (nc < 1) nc = 1;
  if (nc > rp::kScanC) nc = rp::kScanC;
  if (phase == rp::PH_PUSH) {
    s_comp_of[threadIdx.x] = 0xFFu;   // (kRpThreads == 256 buckets at most)
    __syncthreads();
    if ((int)threadIdx.x < nc && threadIdx.x < c.push_n) {
      const uint32_t b = c.push_b[threadIdx.x];
      s_comp_of[b & 0xFFu] = threadIdx.x;
      s_comp_bucket[threadIdx.x] = b;
      s_comp_tail[threadIdx.x] = c.tail[b];
    }
    __syncthreads();
  }
  RpPhaseScan f{a, phase, s_comp_of, s_comp_bucket, s_comp_tail, phase == rp::PH_PUSH && c.push_last != 0};
  rp_scan_tiles(f, sc, n, a.ctl->scan_tot, &a.ctl->error, nc, /*by_block=*/true,
                (a.wg_stats && phase == rp::PH_PUSH && blockIdx.x < 4096u) ? a.wg_stats + (size_t)4096 * (rp::kWgStats + 32) + (size_t)blockIdx.x * 8 : nullptr);
}

__host__ __device__ inline unsigned long long rp_hdr(uint32_t seq, uint32_t phase, uint32_t n) {
  return (unsigned long long)n | (unsigned long long)(phase & 0xFFu) << 32 | (unsigned long long)(seq & (kRpGraphSteps - 1)) << 40;
}
__host__ __device__ inline uint32_t rp_hdr_seq(unsigned long long h) { return (uint32_t)(h >> 40) & 0xFFu; }
__host__ __device__ inline uint32_t rp_hdr_phase(unsigned long long h) { return (uint32_t)(h >> 32) & 0xFFu; }
template <bool SERIAL>
__global__ void __launch_bounds__(kRpThreads) k_rp_step(rp::Args a, RpScan sc, uint32_t seq) {
  __shared__ uint32_t s_last;
  RpLds& s_u = g_rp_lds;
  SimLds& s_sim = s_u.sim;
  // COMMIT_FOLD / RAISE_FOLD: pushes per queue and relaxations of this workgroup.  Ctl::push_cnt and Ctl::st_relax share a
  // few cache lines and used to take one atomic per push and one per target — 15 k on the same lines in a launch over 7 k
  // targets, which is what such a launch lasted (a line takes a few hundred atomics per microsecond at best)
  __shared__ uint32_t s_push[rp::kMaxBuckets + 1];
  __shared__ uint32_t s_relax;
  rp::Ctl& c = *a.ctl;
  // Only the workgroups with work are waited for, so the control step of this launch can run before the dispatcher has
  // started the rest of the grid; those find the header of the next launch and leave.
  // (asked for with the header — part A, as the last launch's control step left it: PH_APPLY's copy of the lists' shard starts,
  // on its way while the header is awaited instead of in a trip of its own behind it)
  uint32_t pre_v = 0;
  if (threadIdx.x < 2 * (rp::kShards + 1)) pre_v = threadIdx.x <= rp::kShards ? c.chg_pre[threadIdx.x] : c.born_pre[threadIdx.x - (rp::kShards + 1)];
  const unsigned long long hdr = __hip_atomic_load(&c.hdr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  if (rp_hdr_seq(hdr) != seq) return;
  const uint32_t phase = rp_hdr_phase(hdr);
  if (phase == rp::PH_DONE) return;
  const uint32_t n = (uint32_t)hdr;
  // workgroups that have something to do (the others leave at once and are not waited for: 512 arrivals on one
  // counter cost 6 us, a phase of a few hundred items should not pay for them)
  const bool per_wave = !SERIAL && (phase == rp::PH_FOLD || phase == rp::PH_COMMIT_FOLD || phase == rp::PH_RAISE_FOLD);
  const uint32_t unit = (!SERIAL && phase == rp::PH_SIM) ? 1 : (per_wave ? kRpThreads / 64 : kRpThreads);
  uint32_t active = (n + unit - 1) / unit;
  if (active > gridDim.x) active = gridDim.x;
  if (active == 0) active = 1;
  if (blockIdx.x >= active) return;
  const bool lds_counts = !SERIAL && a.c.lds_counts && (phase == rp::PH_COMMIT_FOLD || phase == rp::PH_RAISE_FOLD);
  if (lds_counts) {
    for (uint32_t i = threadIdx.x; i <= rp::kMaxBuckets; i += kRpThreads) s_push[i] = 0;
    if (threadIdx.x == 0) s_relax = 0;
    __syncthreads();
  }
  // PH_APPLY and PH_SIM mark dirty targets: collected per workgroup, one range of the list taken at the end
  const bool collect = !SERIAL && phase == rp::PH_APPLY;   // (PH_SIM collects per ranking, inside rp_sim_block)
  if (threadIdx.x == 0) {
    g_rp_collect = collect ? 1u : 0u;
    if (collect) s_u.dirty.cnt = 0;
    g_rp_pre_born = c.born_pre;
  }
  if (threadIdx.x < 2 * (rp::kShards + 1)) g_rp_pre[threadIdx.x] = pre_v;
  __syncthreads();
  if (phase == rp::PH_RANK || phase == rp::PH_PUSH) {
    rp_scan_phase(a, sc, phase == rp::PH_PUSH ? c.scan_n : n, phase);
    if (phase == rp::PH_PUSH && c.push_last) {   // the last pass cleans up behind itself (n: the larger of the two item counts)
      const uint32_t stride = active * kRpThreads;
      for (uint32_t tid = blockIdx.x * kRpThreads + threadIdx.x; tid < n; tid += stride) rp::rp_phase_cleanup(a, tid, /*skip_committed=*/true);
    }
  } else if (!SERIAL && phase == rp::PH_RAISE_FOLD && a.c.fold_pairs && n > gridDim.x * (kRpThreads / 64) && n <= gridDim.x * (kRpThreads / 64) * 128u) {
    // (more targets than waves: two at a time, as in PH_FOLD below)
    const uint32_t wave = threadIdx.x >> 6, waves = active * (kRpThreads / 64);
    const int lane = threadIdx.x & 63;
    uint32_t def_p = 0, n_def = 0;
    for (uint32_t p = blockIdx.x * (kRpThreads / 64) + wave; 2u * p < n; p += waves) {
      const uint32_t w = 2u * p + ((uint32_t)lane >> 5);
      const bool act = w < n;
      const uint32_t gid = act ? a.tgt_gid[w] : rp::kNone;
      uint32_t cnt = act ? a.tgt_cnt[w] : 0u;
      if (cnt > a.c.ev) cnt = a.c.ev;
      if (__ballot(act && gid != rp::kNone && cnt > 32u)) {
        if ((uint32_t)lane == n_def) def_p = p;
        ++n_def;
        continue;
      }
      rp_fold_raise_pair<32>(a, w, gid, cnt, act, lane, lds_counts ? s_push : nullptr);
    }
    for (uint32_t i = 0; i < 2u * n_def; ++i) {
      const uint32_t w = 2u * rl_u32(def_p, (int)(i >> 1)) + (i & 1u);
      if (w < n) rp_fold_raise_wave(a, w, lane, lds_counts ? s_push : nullptr);
    }
  } else if (!SERIAL && phase == rp::PH_RAISE_FOLD) {
    const uint32_t wave = threadIdx.x >> 6, waves = active * (kRpThreads / 64);
    for (uint32_t w = blockIdx.x * (kRpThreads / 64) + wave; w < n; w += waves)
      rp_fold_raise_wave(a, w, threadIdx.x & 63, lds_counts ? s_push : nullptr);
  } else if (!SERIAL && (phase == rp::PH_FOLD || phase == rp::PH_COMMIT_FOLD) && a.c.fold_pairs && n > gridDim.x * (kRpThreads / 64) &&
             n <= gridDim.x * (kRpThreads / 64) * 128u) {
    // more targets than waves: lists of up to 32 events are folded two at a time, a half-wave each (rp_fold_pair).  A pair
    // with a longer list is noted (the i-th such pair in lane i: a wave sees at most 64 pairs) and folded afterwards, one
    // target after the other with the whole wave — apart from the pairs' loop, so that rp_fold_wave's registers are not held
    // on top of it.  (Four targets per wave, 16 lanes and events each, were measured too: too many lists are longer, the
    // groups folded singly after all cost what the quads gain.)
    const uint32_t wave = threadIdx.x >> 6, waves = active * (kRpThreads / 64);
    const int lane = threadIdx.x & 63;
    const bool is_fold = phase == rp::PH_FOLD;
    const uint32_t* dl = (is_fold && !c.fold_all) ? a.dl[c.read] : nullptr;
    constexpr uint32_t per = 2u, gl = 32u;   // targets per wave at a time, lanes (= most events) per target
    uint32_t relax = 0, def_p = 0, n_def = 0;
    {
      const uint32_t h = (uint32_t)lane / gl;
      const unsigned long long limit = is_fold ? rp::kNever : c.cut;
      for (uint32_t p = blockIdx.x * (kRpThreads / 64) + wave; per * p < n; p += waves) {
        const uint32_t w = per * p + h;
        const bool act = w < n;
        const uint32_t t = act ? (dl ? dl[w] : w) : 0u;
        const uint32_t gid = act ? a.tgt_gid[t] : rp::kNone;
        uint32_t cnt = act ? a.tgt_cnt[t] : 0u;
        if (cnt > a.c.ev) cnt = a.c.ev;
        if (__ballot(act && gid != rp::kNone && cnt > gl)) {
          if ((uint32_t)lane == n_def) def_p = p;
          ++n_def;
          continue;
        }
        if (is_fold && act && (uint32_t)lane % gl == 0u) a.tgt_dirty[t] = 0;
        rp_fold_pair<32>(a, t, gid, cnt, act, limit, !is_fold, lane, s_u.fold[wave], lds_counts ? s_push : nullptr, lds_counts ? &relax : nullptr);
      }
      if (!is_fold && (uint32_t)lane % gl == 0u && relax) atomicAdd(&s_relax, relax);
    }
    for (uint32_t i = 0; i < per * n_def; ++i) {
      const uint32_t w = per * rl_u32(def_p, (int)(i / per)) + i % per;
      if (w >= n) continue;
      if (is_fold) {
        const uint32_t t = dl ? dl[w] : w;
        if (lane == 0) a.tgt_dirty[t] = 0;
        rp_fold_wave(a, t, rp::kNever, false, lane, s_u.fold[wave]);
      } else {
        uint32_t rl = 0;
        rp_fold_wave(a, w, c.cut, true, lane, s_u.fold[wave], lds_counts ? s_push : nullptr, lds_counts ? &rl : nullptr);
        if (lane == 0 && rl) atomicAdd(&s_relax, rl);
      }
    }
  } else if (!SERIAL && (phase == rp::PH_FOLD || phase == rp::PH_COMMIT_FOLD)) {
    // one wave per target
    const uint32_t wave = threadIdx.x >> 6, waves = active * (kRpThreads / 64);
    const int lane = threadIdx.x & 63;
    for (uint32_t w = blockIdx.x * (kRpThreads / 64) + wave; w < n; w += waves) {
      if (phase == rp::PH_FOLD) {
        const uint32_t t = c.fold_all ? w : a.dl[c.read][w];
        if (lane == 0) a.tgt_dirty[t] = 0;
        rp_fold_wave(a, t, rp::kNever, false, lane, s_u.fold[wave]);
      } else {
        uint32_t relax = 0;
        rp_fold_wave(a, w, c.cut, true, lane, s_u.fold[wave], lds_counts ? s_push : nullptr, lds_counts ? &relax : nullptr);
        if (lane == 0 && relax) atomicAdd(&s_relax, relax);
      }
    }
  } else if (!SERIAL && phase == rp::PH_SIM) {
    for (uint32_t w = blockIdx.x; w < n; w += active) rp_sim_block(a, a.sd_list[w], s_sim);
  } else {
    const uint32_t stride = active * kRpThreads;
    for (uint32_t tid = blockIdx.x * kRpThreads + threadIdx.x; tid < n; tid += stride) rp_run_phase<SERIAL>(a, phase, tid);
  }
  // the last workgroup to get here picks the next phase.  Every workgroup only has to have its own atomics performed
  // before it says so: the last one reads the control block through atomic read-modify-writes (the per-XCD L2s are not
  // coherent) — all words at once, one per thread, into LDS; rp_control then runs on the LDS copy (a dozen dependent
  // trips to memory otherwise, 10 - 20 us per step) and the copy is stored back.
  if (lds_counts) {
    __syncthreads();
    uint32_t* row = a.push_shards + (size_t)(blockIdx.x % rp::kPushShards) * (rp::kMaxBuckets + 2);   // (Args::push_shards)
    for (uint32_t i = threadIdx.x; i <= rp::kMaxBuckets; i += kRpThreads)
      if (s_push[i]) atomicAdd(&row[i], s_push[i]);
    if (threadIdx.x == 0 && s_relax) atomicAdd(&row[rp::kMaxBuckets + 1], s_relax);
  }
  if (collect) {
    __syncthreads();
    const uint32_t nd = s_u.dirty.cnt < kWgDirtyCap ? s_u.dirty.cnt : kWgDirtyCap;
    const uint32_t wl = 1u - c.read;
    if (threadIdx.x == 0 && nd) s_u.dirty.base = atomicAdd(&c.n_dirty[wl], nd);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nd; i += kRpThreads) a.dl[wl][s_u.dirty.base + i] = s_u.dirty.list[i];   // (< tgt_cap: a target is listed once)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    // the last workgroup to arrive runs the control step.  More than 64 of them arrive in two levels (Ctl::arrive_sub): a line
    // takes ~87 arrivals per microsecond
    uint32_t last;
    if (active == 1u) {
      last = 1u;   // (the only workgroup with work need not ask whether it is the last one)
    } else if (active <= 64u) {
      last = atomicAdd(&c.arrive, 1u) == active - 1 ? 1u : 0u;
    } else {
      const uint32_t sub = blockIdx.x & (rp::kArriveSubs - 1);
      const uint32_t expect = (active - sub + rp::kArriveSubs - 1) / rp::kArriveSubs;   // workgroups sub, sub + 32, ... below `active`
      last = 0;
      if (atomicAdd(&c.arrive_sub[sub].v, 1u) == expect - 1) {
        atomicExch(&c.arrive_sub[sub].v, 0u);
        last = atomicAdd(&c.arrive, 1u) == rp::kArriveSubs - 1 ? 1u : 0u;
      }
    }
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  const unsigned long long t_arrive = a.c.stats ? wall_clock64() : 0ull;
  __syncthreads();   // (everybody is done with the phase's part of the shared region)
  rp::Ctl& s_ctl = s_u.ctl;
  // What the control step needs of the block, a word per thread and ONE trip: part A, the words in use of part B's lines, the
  // first num_buckets + 1 entries of the five per-queue arrays.  (Until round 6 all of it, statistics and — once the counters
  // had a line each — padding included: 725 words read with atomics and stored back; a launch pays ~1 us per 128 of them.)
  // The statistics are not loaded: zeros in the LDS copy, and what the step counted is ADDED to the block afterwards.
  constexpr uint32_t kL = 5, kB = 6 + 2 * kL + rp::kTgtShards + 2 * rp::kShards;
  constexpr uint32_t kA = offsetof(rp::Ctl, error) / 4;
  const uint32_t nq = (uint32_t)a.c.num_buckets + 1u, n_copy = kA + kB + 5u * nq;
  const auto ctl_word = [&](uint32_t i) -> uint32_t {
    if (i < kA) return i;
    i -= kA;
    if (i < 6) return (uint32_t)(offsetof(rp::Ctl, error) / 4) + i;   // error, k_limit, first_change, smax_cut
    if (i < kB) {
      constexpr uint32_t line[kL] = {offsetof(rp::Ctl, n_tgt) / 4, offsetof(rp::Ctl, n_dirty) / 4, offsetof(rp::Ctl, n_sd) / 4, offsetof(rp::Ctl, n_cp) / 4, offsetof(rp::Ctl, arrive) / 4};
      constexpr uint32_t kW = sizeof(rp::CtlLine) / 4;
      if (i < 6 + 2 * kL) return line[(i - 6) >> 1] + ((i - 6) & 1u);
      i -= 6 + 2 * kL;
      if (i < rp::kTgtShards) return (uint32_t)(offsetof(rp::Ctl, tgt_n) / 4) + i * kW;
      i -= rp::kTgtShards;
      if (i < rp::kShards) return (uint32_t)(offsetof(rp::Ctl, chg_n) / 4) + i * kW;
      return (uint32_t)(offsetof(rp::Ctl, born_n) / 4) + (i - rp::kShards) * kW;
    }
    i -= kB;
    return (uint32_t)(offsetof(rp::Ctl, head) / 4) + (i / nq) * (rp::kMaxBuckets + 1) + i % nq;
  };
  constexpr uint32_t kStat0 = offsetof(rp::Ctl, st_raise_pops) / 8, kStat1 = sizeof(rp::Ctl) / 8;
  {
    uint32_t* src = reinterpret_cast<uint32_t*>(a.ctl);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&s_ctl);
    for (uint32_t i0 = 0; i0 < n_copy; i0 += 2 * kRpThreads) {   // (one trip: n_copy <= 512 with the reference's 20 buckets)
      uint32_t v[2], wv[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const uint32_t i = i0 + k * kRpThreads + threadIdx.x;
        wv[k] = ctl_word(i < n_copy ? i : 0u);
        v[k] = i < n_copy ? atomicAdd(&src[wv[k]], 0u) : 0u;
      }
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (i0 + k * kRpThreads + threadIdx.x < n_copy) dst[wv[k]] = v[k];
    }
    unsigned long long* st = reinterpret_cast<unsigned long long*>(&s_ctl);
    for (uint32_t i = kStat0 + threadIdx.x; i < kStat1; i += kRpThreads) st[i] = 0ull;
  }
  __syncthreads();
  if (lds_counts) {   // the workgroups' push counts and relaxations: summed out of their shards, which are left at zero
    const uint32_t i = threadIdx.x <= (uint32_t)a.c.num_buckets ? threadIdx.x : (threadIdx.x == kRpThreads - 1 ? rp::kMaxBuckets + 1 : ~0u);
    if (i != ~0u) {
      uint32_t v[rp::kPushShards], sum = 0;
#pragma unroll
      for (uint32_t sh = 0; sh < rp::kPushShards; ++sh) v[sh] = atomicExch(&a.push_shards[(size_t)sh * (rp::kMaxBuckets + 2) + i], 0u);
#pragma unroll
      for (uint32_t sh = 0; sh < rp::kPushShards; ++sh) sum += v[sh];
      if (i <= rp::kMaxBuckets) s_ctl.push_cnt[i] += sum;
      else s_ctl.st_relax += sum;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    s_ctl.arrive = 0;
    const unsigned long long now = a.c.stats ? wall_clock64() : 0ull;   // 100 MHz
    if (a.c.stats && s_ctl.t_prev) {
      s_ctl.st_phase_ticks[phase & 15] += now - s_ctl.t_prev;
      const int row = phase == rp::PH_FOLD ? 0 : phase == rp::PH_APPLY ? 1 : phase == rp::PH_SIM ? 2 : phase == rp::PH_PLACE_BASE ? 3 : phase == rp::PH_PUSH ? 4 : phase == rp::PH_COMMIT_FOLD ? 5 : phase == rp::PH_CLEANUP ? 6 : phase == rp::PH_RAISE_FOLD ? 7 : -1;
      if (row >= 0) {
        int bin = 0;
        for (uint32_t lim = 4; bin < 7 && n >= lim; lim <<= 2) ++bin;
        s_ctl.st_bin_steps[row][bin] += 1;
        s_ctl.st_bin_ticks[row][bin] += now - s_ctl.t_prev;
      }
    }
    if (a.c.stats) s_ctl.t_prev = now;
    if (phase == rp::PH_RANK || phase == rp::PH_PUSH) {
      sc.ticket[0] = 0;
      sc.ticket[1] = sc.ticket[1] + 1;
    }
    rp::Args a2 = a;
    a2.ctl = &s_ctl;
    const unsigned long long tc0 = a.c.stats ? wall_clock64() : 0ull;
    rp::rp_control(a2);
    if (a.c.stats) {
      const unsigned long long tc1 = wall_clock64();
      s_ctl.st_ctl_ticks[0] += now - t_arrive;   // arrival of the last workgroup -> copy in done
      s_ctl.st_ctl_ticks[1] += tc1 - tc0;        // rp_control
      s_ctl.st_ctl_ticks[2] += tc0 - now;        // statistics in front of it
    }
    s_ctl.hdr = rp_hdr(seq + 1, s_ctl.phase, s_ctl.n_threads);
  }
  __syncthreads();
  {
    uint32_t* dst = reinterpret_cast<uint32_t*>(a.ctl);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&s_ctl);
    const uint32_t w_hdr = offsetof(rp::Ctl, hdr) / 4;
    for (uint32_t i = threadIdx.x; i < n_copy; i += kRpThreads) {
      const uint32_t w = ctl_word(i);
      if (w != w_hdr && w != w_hdr + 1) dst[w] = src[w];
    }
    unsigned long long* gst = reinterpret_cast<unsigned long long*>(a.ctl);
    const unsigned long long* st = reinterpret_cast<const unsigned long long*>(&s_ctl);
    for (uint32_t i = kStat0 + threadIdx.x; i < kStat1; i += kRpThreads)
      if (st[i]) atomicAdd(&gst[i], st[i]);
  }
  // the header in one piece (a workgroup that is late only ever reads this word)
  if (threadIdx.x == 0) *reinterpret_cast<volatile unsigned long long*>(&a.ctl->hdr) = s_ctl.hdr;
}

// first control step of an update (PH_BEGIN), one thread
__global__ void k_rp_begin(rp::Args a) {
  a.ctl->phase = rp::PH_BEGIN;
  a.ctl->done = 0;
  rp::rp_control(a);
  a.ctl->hdr = rp_hdr(0, a.ctl->phase, a.ctl->n_threads);
}

}  // namespace
