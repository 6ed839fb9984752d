// vbx_kernels_esdf_replay.hpp — device side of the parallel reference-order open set (vbx_esdf_replay_core.hpp)
// Part of libvbx_hip.so's single translation unit (included by vbx_hip.hip, in order).
//
// k_rp_step runs ONE phase of the replay over the whole grid and lets the last workgroup to finish run rp_control,
// which writes the next phase into the control block.  The host launches the same kernel back to back (a captured
// graph of kRpGraphSteps launches) and looks at Ctl::done now and then: a kernel boundary is the grid-wide barrier
// (1.5 - 2 us, MI355X_MICROARCH.md "boundary"), every phase gets the whole chip, no workgroup ever waits for
// another inside a launch except in the two SCAN phases, where a tile waits for tiles with smaller tickets only
// (chained scan with decoupled look-back: those tiles are running or done, so the wait ends).

#define RP_FN __device__ inline
#define RP_LD(x) atomicAdd(&(x), 0u)
#define RP_LD64(x) atomicAdd(&(x), 0ull)
#include "vbx_esdf_replay_core.hpp"

namespace {

constexpr int kRpThreads = 256;
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

__device__ inline void rp_run_phase(const rp::Args& a, uint32_t phase, uint32_t tid) {
  switch (phase) {
    case rp::PH_PLACE_BASE: rp::rp_phase_place_base(a, tid); break;
    case rp::PH_FOLD: rp::rp_phase_fold(a, tid); break;
    case rp::PH_APPLY: rp::rp_phase_apply(a, tid); break;
    case rp::PH_SIM: rp::rp_phase_sim(a, tid); break;
    case rp::PH_MINCUT: rp::rp_phase_mincut(a, tid); break;
    case rp::PH_COMMIT_FOLD: rp::rp_phase_commit_fold(a, tid); break;
    case rp::PH_RANK_WRITE: rp::rp_phase_rank_write(a, tid); break;
    case rp::PH_CLEANUP: rp::rp_phase_cleanup(a, tid); break;
    default: break;
  }
}

// SCAN phase: tiles of kRpThreads items by ticket; exclusive prefix of the four counts per item
__device__ inline void rp_scan_phase(const rp::Args& a, const RpScan& sc, uint32_t n) {
  __shared__ uint32_t s_tile;
  __shared__ uint32_t s_wave[kRpThreads / 64][4];
  __shared__ uint32_t s_prefix[4];
  rp::Ctl& c = *a.ctl;
  const uint32_t tiles = (n + kRpThreads - 1) / kRpThreads;
  const uint32_t gen = sc.ticket[1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) s_tile = atomicAdd(&sc.ticket[0], 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    if (tile >= tiles) break;
    const uint32_t i = tile * kRpThreads + threadIdx.x;
    rp::Cnt4 cnt = {{0, 0, 0, 0}};
    if (i < n) cnt = rp::rp_scan_count(a, i);
    // exclusive scan inside the tile: wave scan by shuffles, wave totals through LDS
    rp::Cnt4 ex;
    for (int k = 0; k < 4; ++k) {
      uint32_t v = cnt.v[k];
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(v, d);
        if (lane >= d) v += o;
      }
      if (lane == 63) s_wave[wave][k] = v;
      ex.v[k] = v - cnt.v[k];
    }
    __syncthreads();
    uint32_t agg[4];
    for (int k = 0; k < 4; ++k) {
      uint32_t before = 0, total = 0;
      for (int w = 0; w < kRpThreads / 64; ++w) {
        if (w < wave) before += s_wave[w][k];
        total += s_wave[w][k];
      }
      ex.v[k] += before;
      agg[k] = total;
    }
    // publish the aggregate, look back (one lane per component)
    if (threadIdx.x < 4) {
      const int k = threadIdx.x;
      unsigned long long* d = sc.desc + (size_t)tile * 4 + k;
      uint32_t prefix = 0;
      if (tile == 0) {
        __hip_atomic_store(d, ((unsigned long long)((gen << 2) | 2u) << 32) | agg[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        __hip_atomic_store(d, ((unsigned long long)((gen << 2) | 1u) << 32) | agg[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t t = tile;
        uint32_t spins = 0;
        while (t > 0) {
          const unsigned long long w = __hip_atomic_load(sc.desc + (size_t)(t - 1) * 4 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint32_t tag = (uint32_t)(w >> 32);
          if ((tag >> 2) != gen || (tag & 3u) == 0u) {
            if (++spins > kRpSpinMax) { atomicOr(&c.error, 64u); break; }
            __builtin_amdgcn_s_sleep(1);
            continue;
          }
          prefix += (uint32_t)w;
          if ((tag & 3u) == 2u) break;
          --t;
        }
        __hip_atomic_store(d, ((unsigned long long)((gen << 2) | 2u) << 32) | (prefix + agg[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      s_prefix[k] = prefix;
      if (tile == tiles - 1) atomicExch(&c.scan_tot[k], prefix + agg[k]);
    }
    __syncthreads();
    if (i < n) {
      for (int k = 0; k < 4; ++k) ex.v[k] += s_prefix[k];
      rp::rp_scan_apply(a, i, ex);
    }
  }
  if (n == 0 && blockIdx.x == 0 && threadIdx.x < 4) atomicExch(&c.scan_tot[threadIdx.x], 0u);
}

__global__ void __launch_bounds__(kRpThreads) k_rp_step(rp::Args a, RpScan sc) {
  __shared__ uint32_t s_last;
  rp::Ctl& c = *a.ctl;
  const uint32_t phase = c.phase;
  if (phase == rp::PH_DONE) return;
  const uint32_t n = c.n_threads;
  if (phase == rp::PH_RANK || phase == rp::PH_PUSH) {
    rp_scan_phase(a, sc, n);
  } else {
    const uint32_t stride = gridDim.x * kRpThreads;
    for (uint32_t tid = blockIdx.x * kRpThreads + threadIdx.x; tid < n; tid += stride) rp_run_phase(a, phase, tid);
  }
  // the last workgroup to get here picks the next phase (its reads of what the others counted are atomic
  // read-modify-writes, so every workgroup only has to have its own atomics performed before it says so)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) s_last = (atomicAdd(&c.arrive, 1u) == gridDim.x - 1) ? 1u : 0u;
  __syncthreads();
  if (s_last && threadIdx.x == 0) {
    c.arrive = 0;
    if (phase == rp::PH_RANK || phase == rp::PH_PUSH) {
      sc.ticket[0] = 0;
      sc.ticket[1] = sc.ticket[1] + 1;
    }
    rp::rp_control(a);
  }
}

// first control step of an update (PH_BEGIN), one thread
__global__ void k_rp_begin(rp::Args a) {
  a.ctl->phase = rp::PH_BEGIN;
  a.ctl->done = 0;
  rp::rp_control(a);
}

}  // namespace
