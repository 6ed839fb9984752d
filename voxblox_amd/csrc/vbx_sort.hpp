// vbx_sort.hpp — stable LSD radix sort of 64-bit keys (optionally with 32-bit values) on a bit field.
// Part of libvbx_hip.so's single translation unit (included by vbx_hip.hip, in order).
//
// Every sort of the hot path orders 3e5..1e6 keys by a 20..26-bit field and must be STABLE (the
// inputs are produced in visiting order).  rocPRIM's onesweep handles that size with 8-bit
// digits — 3-4 passes, each a launch plus two or three small memsets plus a histogram pass:
// ~105 us for a 20-bit field, all of it launch/latency.  Here a pass takes a digit of up to 12
// bits (20..24-bit fields in TWO passes) and consists of exactly three launches:
//   k_rsort_count    per-workgroup digit histogram (LDS atomics)      -> hist[digit][workgroup]
//   exclusive scan   over hist in digit-major order (k_scan_excl)      -> global base per (digit, workgroup)
//   k_rsort_scatter  stable placement: a workgroup's 2048 keys are split over its 4 waves in
//                    memory order; per wave a running counter per digit lives in LDS, and inside
//                    a 64-key chunk the rank among equal digits comes from ballots (wave64
//                    multisplit: one ballot per digit bit builds the mask of equal-digit lanes).
// Wave order, chunk order and lane order are all memory order, hence stable.
namespace {

constexpr int kSortThreads = 256;                 // 4 waves
constexpr int kSortItems = 8;                     // keys per thread
constexpr int kSortTile = kSortThreads * kSortItems;  // 2048 keys per workgroup
constexpr int kSortMaxBits = 12;

template <int BITS>
__global__ void __launch_bounds__(kSortThreads)
k_rsort_count(const uint64_t* __restrict__ keys, uint32_t n, const uint32_t* __restrict__ n_dev, int shift,
              uint32_t* __restrict__ hist, uint32_t nwg) {
  constexpr int NB = 1 << BITS;
  if (n_dev) n = min(n, *n_dev);  // device-side size (the grid and `hist` are laid out for the host's bound n)
  __shared__ uint32_t s_cnt[NB];
  for (int i = threadIdx.x; i < NB; i += kSortThreads) s_cnt[i] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * kSortTile;
  // loads first, unconditionally (index 0 stands in beyond n): a load inside the `if` with its LDS atomic
  // compiles to load - wait - atomic per item, eight dependent memory round trips instead of one
  uint64_t kv[kSortItems];
#pragma unroll
  for (int e = 0; e < kSortItems; ++e) {
    const uint32_t i = base + e * kSortThreads + threadIdx.x;  // order is irrelevant for counting
    kv[e] = keys[(i < n) ? i : 0u];
  }
#pragma unroll
  for (int e = 0; e < kSortItems; ++e) {
    const uint32_t i = base + e * kSortThreads + threadIdx.x;
    if (i < n) atomicAdd(&s_cnt[(uint32_t)(kv[e] >> shift) & (NB - 1)], 1u);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < NB; d += kSortThreads) hist[(size_t)d * nwg + blockIdx.x] = s_cnt[d];
}

template <int BITS, bool kHasVals>
__global__ void __launch_bounds__(kSortThreads)
k_rsort_scatter(const uint64_t* __restrict__ kin, const uint32_t* __restrict__ vin, uint64_t* __restrict__ kout,
                uint32_t* __restrict__ vout, uint32_t n, const uint32_t* __restrict__ n_dev, int shift,
                const uint32_t* __restrict__ gofs, uint32_t nwg) {
  constexpr int NB = 1 << BITS;
  if (n_dev) n = min(n, *n_dev);
  constexpr int NW = kSortThreads / 64;
  __shared__ uint32_t s_run[NW][NB];  // phase 1: per-wave digit counts; phase 3: start of the wave's next chunk per digit
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < NW * NB; i += kSortThreads) (&s_run[0][0])[i] = 0;
  __syncthreads();
  // a wave owns 64 * kSortItems consecutive keys; its chunk c is keys [wbase + 64 c, wbase + 64 c + 64)
  const uint32_t wbase = blockIdx.x * kSortTile + w * (64 * kSortItems);
  uint64_t key[kSortItems];
  uint32_t dig[kSortItems];
#pragma unroll
  for (int c = 0; c < kSortItems; ++c) {  // all loads in flight before the first LDS atomic (see k_rsort_count)
    const uint32_t i = wbase + c * 64 + lane;
    key[c] = kin[(i < n) ? i : 0u];
  }
#pragma unroll
  for (int c = 0; c < kSortItems; ++c) {
    const uint32_t i = wbase + c * 64 + lane;
    if (i >= n) key[c] = 0;
    dig[c] = (uint32_t)(key[c] >> shift) & (NB - 1);
    if (i < n) atomicAdd(&s_run[w][dig[c]], 1u);
  }
  __syncthreads();
  // per digit: exclusive prefix over the waves, plus this workgroup's global base
  for (int d = threadIdx.x; d < NB; d += kSortThreads) {
    uint32_t acc = gofs[(size_t)d * nwg + blockIdx.x];
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) {
      const uint32_t cnt = s_run[ww][d];
      s_run[ww][d] = acc;
      acc += cnt;
    }
  }
  __syncthreads();
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int c = 0; c < kSortItems; ++c) {
    const uint32_t i = wbase + c * 64 + lane;
    const bool act = i < n;
    // lanes of this chunk holding the same digit
    unsigned long long mask = __ballot(act);
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
      const unsigned long long m = __ballot((dig[c] >> b) & 1u);
      mask &= ((dig[c] >> b) & 1u) ? m : ~m;
    }
    if (act) {
      const uint32_t rank = (uint32_t)__popcll(mask & lt);
      const uint32_t start = s_run[w][dig[c]];  // every lane of the group reads before its leader writes
      const uint32_t pos = start + rank;
      kout[pos] = key[c];
      if (kHasVals) vout[pos] = vin[i];
      if (rank == 0) s_run[w][dig[c]] = start + (uint32_t)__popcll(mask);
    }
  }
}

// ---------------------------------------------------------------------------
// Fused pass (the default up to 4 M keys and 30-bit fields; VBX_SORT_FUSED=0 selects the three launches
// above; see stable_sort01): count, prefix and scatter of
// one digit in ONE launch, after one histogram launch per sort — three launches for a 20-bit field
// instead of six.  A small kernel of this path lasts 8-12 us of which its waves run 2-3 (dispatch ramp and
// the write-back / invalidate at its boundaries are the rest, profiles/r02_pmc_fast_instructions.txt), and
// the Fast integrator sorts seven times per frame.
//   k_rsort_hist   digit histograms of ALL passes in one read of the keys -> hist[pass][digit] (global
//                  atomics into a slot of a ring the host keeps zeroed)
//   k_rsort_fused  tiles of 8192 keys, taken by ticket (a tile only waits for tiles that already run, so
//                  no residency assumption).  A tile counts its digits per wave in LDS, publishes the
//                  tile's counts (16 bit each, write-through stores, drained, then ONE flag word tagged
//                  with the call's generation: the per-XCD L2s are not coherent), waits for the flags of
//                  all earlier tiles, sums their count rows (8-byte agent-scope loads, eight in flight per
//                  lane, four row groups across the workgroup) and scatters exactly like k_rsort_scatter.
// Beyond 512 tiles (4 M keys: the Simple integrator's 35-43 M updates, whole-frame replay rounds at fine voxels) the
// tiles form GROUPS of 512: a tile looks back over the rows of its own group only, and the last tile of a group
// publishes the group's running totals per digit (32 bit) behind a flag of its own, which the tiles of the next group
// add.  Every wait is for a tile with an earlier ticket, i.e. one that is running or done.
// Every spin is bounded: a tile that gives up raises DevState::error bit 64 and the call fails loudly (the passes
// ping-pong between two buffers, so there is no input left to run a slower form on).
// ---------------------------------------------------------------------------
constexpr int kFsThreads = 1024;
constexpr int kFsWaves = kFsThreads / 64;
#ifndef KFS_ITEMS
#define KFS_ITEMS 8
#endif
constexpr int kFsItems = KFS_ITEMS;
constexpr int kFsTile = kFsThreads * kFsItems;  // 8192 keys per workgroup
constexpr int kFsGroup = 512;                   // tiles per look-back group (4 M keys) — a sort of up to one group is one group
constexpr int kFsGroupBig = 64;                 // ... of a sort with more tiles: a tile sums at most 63 count rows (the rows of
                                                // 511 predecessors were 4 x the key traffic of a 37 M key pass)
constexpr int kFsMaxBits = 10;                  // one digit per thread
constexpr int kFsMaxPasses = 3;
constexpr uint32_t kFsSpinMax = 1u << 20;         // polls of an agent-scope load (>= 0.5 us each) + s_sleep: more than half a second
                                                // of waiting for a tile that holds an EARLIER ticket, i.e. one that is resident or
                                                // done — a hang detector, not a scheduling race (a preempted queue comes back in ms)
struct FsPasses {
  int np;
  int shift[kFsMaxPasses];
  uint32_t mask[kFsMaxPasses];
};

__global__ void __launch_bounds__(kFsThreads)
k_rsort_hist(const uint64_t* __restrict__ keys, uint32_t n, const uint32_t* __restrict__ n_dev, FsPasses ps,
             uint32_t* hist) {
  if (n_dev) n = min(n, *n_dev);
  __shared__ uint32_t s_h[kFsMaxPasses][1 << kFsMaxBits];
  for (int i = threadIdx.x; i < kFsMaxPasses * (1 << kFsMaxBits); i += kFsThreads) (&s_h[0][0])[i] = 0;
  __syncthreads();
  // (a workgroup takes every gridDim.x-th tile: the host caps the grid, so that a sort of tens of millions of keys
  // ends in a few hundred flushes of the LDS counters instead of thousands)
  for (uint32_t tile = blockIdx.x; (unsigned long long)tile * kFsTile < n; tile += gridDim.x) {
    const uint32_t base = tile * kFsTile;
    uint64_t kv[kFsItems];
#pragma unroll
    for (int e = 0; e < kFsItems; ++e) {  // all loads in flight before the first LDS atomic (see k_rsort_count)
      const uint32_t i = base + e * kFsThreads + threadIdx.x;
      kv[e] = keys[(i < n) ? i : 0u];
    }
#pragma unroll
    for (int e = 0; e < kFsItems; ++e) {
      const uint32_t i = base + e * kFsThreads + threadIdx.x;
      if (i < n)
        for (int p = 0; p < ps.np; ++p) atomicAdd(&s_h[p][(uint32_t)(kv[e] >> ps.shift[p]) & ps.mask[p]], 1u);
    }
  }
  __syncthreads();
  for (int p = 0; p < ps.np; ++p)
    for (uint32_t d = threadIdx.x; d <= ps.mask[p]; d += kFsThreads)
      if (s_h[p][d]) atomicAdd(&hist[p * (1 << kFsMaxBits) + d], s_h[p][d]);
}

#ifdef VBX_SORT_STATS  // measurement build (tools/sort_stats.py): where a tile of the fused pass spends its time
__device__ unsigned long long g_sort_stats[16];
#define SORT_T(i) if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&g_sort_stats[i], t_ - t_prev); atomicMax(&g_sort_stats[8 + i], t_ - t_prev); t_prev = t_; }
#else
#define SORT_T(i)
#endif
template <int BITS, bool kHasVals>
__global__ void __launch_bounds__(kFsThreads)
k_rsort_fused(const uint64_t* __restrict__ kin, const uint32_t* __restrict__ vin, uint64_t* __restrict__ kout,
              uint32_t* __restrict__ vout, uint32_t n, const uint32_t* __restrict__ n_dev, int shift,
              const uint32_t* __restrict__ hist, uint16_t* cnt16, unsigned long long* flags, uint32_t* ticket,
              uint32_t ticket_base, uint32_t gen, DevState* st, unsigned long long* gflags, uint32_t* gpre, uint32_t gsize) {
  constexpr int NB = 1 << BITS;
  constexpr int WORDS = NB / 4;             // 8-byte words per count row
  constexpr int GROUPS = kFsThreads / WORDS;  // row groups of the look-back
  static_assert(BITS >= 4 && BITS <= kFsMaxBits, "one digit per thread, at least one word per row");
  if (n_dev) n = min(n, *n_dev);
  __shared__ uint32_t s_run[kFsWaves][NB];  // per-wave digit counts, then the start of the wave's next chunk per digit
  __shared__ uint32_t s_base[NB];           // global start of the digit + keys of earlier tiles
  __shared__ __attribute__((aligned(8))) uint16_t s_c16[NB];
  __shared__ uint32_t s_wsum[kFsWaves];
  __shared__ uint32_t s_tile;
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
#ifdef VBX_SORT_STATS
  unsigned long long t_prev = wall_clock64();
  if (threadIdx.x == 0) atomicAdd(&g_sort_stats[7], 1ull);
#endif
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u) - ticket_base;
  for (int i = threadIdx.x; i < kFsWaves * NB; i += kFsThreads) (&s_run[0][0])[i] = 0;
  // exclusive scan of the digit histogram -> start of every digit in the output
  const uint32_t hv = (threadIdx.x < NB) ? hist[threadIdx.x] : 0u;
  uint32_t hincl = hv;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(hincl, d);
    if (lane >= d) hincl += o;
  }
  if (lane == 63) s_wsum[w] = hincl;
  __syncthreads();
  const uint32_t tile = s_tile;
  if ((unsigned long long)tile * kFsTile >= n) return;  // uniform: an empty tile has no successors that need it
  const uint32_t group = tile / gsize, local = tile % gsize, gfirst = group * gsize;
  uint32_t hstart = 0;  // start of this thread's digit in the output
  {
    uint32_t wb = 0;
#pragma unroll
    for (int ww = 0; ww < kFsWaves; ++ww)
      if (ww < w) wb += s_wsum[ww];
    hstart = wb + hincl - hv;
    if (threadIdx.x < NB) s_base[threadIdx.x] = hstart;
  }
  // a wave owns 64 * kFsItems consecutive keys; its chunk c is keys [wbase + 64 c, wbase + 64 c + 64)
  const uint32_t wbase = tile * kFsTile + w * (64 * kFsItems);
  uint64_t key[kFsItems];
  uint32_t dig[kFsItems];
#pragma unroll
  for (int c = 0; c < kFsItems; ++c) {  // all loads in flight before the first LDS atomic (see k_rsort_count)
    const uint32_t i = wbase + c * 64 + lane;
    key[c] = kin[(i < n) ? i : 0u];
  }
#pragma unroll
  for (int c = 0; c < kFsItems; ++c) {
    const uint32_t i = wbase + c * 64 + lane;
    if (i >= n) key[c] = 0;
    dig[c] = (uint32_t)(key[c] >> shift) & (NB - 1);
    if (i < n) atomicAdd(&s_run[w][dig[c]], 1u);
  }
  __syncthreads();
  SORT_T(0)  // ticket, LDS clear, histogram scan, key loads, per-wave counting
  if (threadIdx.x < NB) {  // exclusive prefix over the waves; the tile's count of the digit
    uint32_t acc = 0;
#pragma unroll
    for (int ww = 0; ww < kFsWaves; ++ww) {
      const uint32_t cnt = s_run[ww][threadIdx.x];
      s_run[ww][threadIdx.x] = acc;
      acc += cnt;
    }
    s_c16[threadIdx.x] = (uint16_t)acc;  // <= 8192
  }
  __syncthreads();
  // publish: the count row write-through, every writing wave drained, then the flag
  unsigned long long* row = reinterpret_cast<unsigned long long*>(cnt16 + (size_t)tile * NB);
  if (threadIdx.x < WORDS)
    __hip_atomic_store(row + threadIdx.x, reinterpret_cast<const unsigned long long*>(s_c16)[threadIdx.x],
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned long long want = ((unsigned long long)gen << 32) | 1ull;
  if (threadIdx.x == 0) __hip_atomic_store(&flags[tile], want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  SORT_T(1)  // wave prefix, row publish + drain, flag
  // earlier tiles of the group: thread t waits for the group's tile t; the last thread for the group before
  bool gave_up = false;
  {
    const unsigned long long* f = nullptr;
    if (threadIdx.x < local) f = &flags[gfirst + threadIdx.x];
    else if (group > 0 && threadIdx.x == kFsThreads - 1) f = &gflags[group - 1];
    if (f) {
      uint32_t spins = 0;
      while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
        if (++spins > kFsSpinMax) {
          gave_up = true;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
  }
  if (__syncthreads_or(gave_up ? 1 : 0)) {
    if (threadIdx.x == 0) atomicOr(&st->error, 64u);
    return;
  }
  SORT_T(2)  // waiting for the flags
  // sum of the earlier tiles' rows: word column q, row group g; eight loads in flight per lane
  {
    const int q = threadIdx.x % WORDS, g = threadIdx.x / WORDS;
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    const unsigned long long* col = reinterpret_cast<const unsigned long long*>(cnt16) + q;
    for (uint32_t r0 = gfirst + g; r0 < tile; r0 += 8 * GROUPS) {
      unsigned long long x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {  // unconditional loads (row r0 stands in beyond the tile), masked below
        const uint32_t r = r0 + j * GROUPS;
        x[j] = __hip_atomic_load(col + (size_t)((r < tile) ? r : r0) * WORDS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (r0 + j * GROUPS >= tile) x[j] = 0ull;
        a0 += (uint32_t)(x[j] & 0xFFFFu);
        a1 += (uint32_t)((x[j] >> 16) & 0xFFFFu);
        a2 += (uint32_t)((x[j] >> 32) & 0xFFFFu);
        a3 += (uint32_t)(x[j] >> 48);
      }
    }
    if (a0) atomicAdd(&s_base[4 * q + 0], a0);
    if (a1) atomicAdd(&s_base[4 * q + 1], a1);
    if (a2) atomicAdd(&s_base[4 * q + 2], a2);
    if (a3) atomicAdd(&s_base[4 * q + 3], a3);
  }
  __syncthreads();
  if (threadIdx.x < NB) {
    uint32_t b = s_base[threadIdx.x];
    if (group > 0)  // keys of the groups before
      b += __hip_atomic_load(&gpre[(size_t)(group - 1) * (1 << kFsMaxBits) + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (local == gsize - 1)  // the group's last tile: totals up to and including this tile, for the next group
      __hip_atomic_store(&gpre[(size_t)group * (1 << kFsMaxBits) + threadIdx.x], b - hstart + (uint32_t)s_c16[threadIdx.x],
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int ww = 0; ww < kFsWaves; ++ww) s_run[ww][threadIdx.x] += b;
  }
  if (local == gsize - 1) {  // uniform
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&gflags[group], want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  SORT_T(3)  // row sums, bases
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int c = 0; c < kFsItems; ++c) {
    const uint32_t i = wbase + c * 64 + lane;
    const bool act = i < n;
    unsigned long long mask = __ballot(act);  // lanes of this chunk holding the same digit
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
      const unsigned long long m = __ballot((dig[c] >> b) & 1u);
      mask &= ((dig[c] >> b) & 1u) ? m : ~m;
    }
    if (act) {
      const uint32_t rank = (uint32_t)__popcll(mask & lt);
      const uint32_t start = s_run[w][dig[c]];  // every lane of the group reads before its leader writes
      const uint32_t pos = start + rank;
      kout[pos] = key[c];
      if (kHasVals) vout[pos] = vin[i];
      if (rank == 0) s_run[w][dig[c]] = start + (uint32_t)__popcll(mask);
    }
  }
#ifdef VBX_SORT_STATS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#endif
  SORT_T(4)  // ranking and scatter (stores drained in the measurement build)
}

// ---------------------------------------------------------------------------
// Exclusive prefix sum of uint32, ONE launch (chained scan with decoupled look-back).
// The path scans 10^5..10^6 counters eight to forty times per frame (ray offsets, compaction, the
// digit histograms of every radix pass, the replay's probe offsets); rocPRIM's scan is two launches
// (look-back state init + scan) plus its host-side dispatch, ~10-12 us per call, i.e. 0.2-0.5 ms of
// a 1.4 ms frame.  Here: a workgroup takes a ticket (tiles are processed in ticket order, so a tile
// only ever waits for tiles that are already running), scans its 4096 items in registers/LDS,
// publishes {generation, status, value} as ONE 64-bit word (aggregate first, inclusive prefix once
// known) with agent-scope atomics — the per-XCD L2s are not coherent, a plain store would not be seen
// — and wave 0 looks back over the predecessors' words, 64 at a time.  The generation tag makes the
// descriptor array reusable without clearing: words of earlier calls read as "not there yet".
// ---------------------------------------------------------------------------
constexpr int kScanThreads = 256;
constexpr int kScanItems = 16;
constexpr int kScanTile = kScanThreads * kScanItems;  // 4096
constexpr unsigned long long kScanAgg = 1ull << 62, kScanIncl = 2ull << 62;

__global__ void __launch_bounds__(kScanThreads)
k_scan_excl(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n, unsigned long long* desc,
            uint32_t* ticket, uint32_t ticket_base, uint32_t gen) {
  __shared__ uint32_t s_tile, s_prefix;
  __shared__ uint32_t s_wsum[kScanThreads / 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u) - ticket_base;
  __syncthreads();
  const uint32_t tile = s_tile;
  const uint32_t base = tile * kScanTile + threadIdx.x * kScanItems;
  uint32_t v[kScanItems];
  uint32_t tsum = 0;
#pragma unroll
  for (int e = 0; e < kScanItems; ++e) {
    v[e] = (base + e < n) ? in[base + e] : 0u;
    tsum += v[e];
  }
  // inclusive scan of the thread sums inside the wave
  uint32_t incl = tsum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) s_wsum[wv] = incl;
  __syncthreads();
  uint32_t wbase = 0, agg = 0;
#pragma unroll
  for (int w = 0; w < kScanThreads / 64; ++w) {
    if (w < wv) wbase += s_wsum[w];
    agg += s_wsum[w];
  }
  const unsigned long long tag = (unsigned long long)(gen & 0x3FFFFFFFu) << 32;
  if (wv == 0) {
    uint32_t excl = 0;
    if (tile > 0) {
      if (lane == 0)
        __hip_atomic_store(&desc[tile], kScanAgg | tag | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // look back: lane l reads the word of tile (first - l), nearest predecessor in lane 0
      int first = (int)tile - 1;
      for (;;) {
        const int idx = first - lane;
        unsigned long long d = kScanIncl | tag;  // virtual tiles before the first: prefix 0
        if (idx >= 0) {
          do {
            d = __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } while ((d & (0x3FFFFFFFull << 32)) != tag || (d >> 62) == 0);
        }
        const unsigned long long has_incl = __ballot((d >> 62) == 2);
        const int stop = has_incl ? (__ffsll((long long)has_incl) - 1) : 64;  // nearest tile with a full prefix
        uint32_t part = (lane <= stop) ? (uint32_t)d : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        excl += part;
        if (has_incl) break;
        first -= 64;
      }
    }
    if (lane == 0) {
      __hip_atomic_store(&desc[tile], kScanIncl | tag | (unsigned long long)(excl + agg), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      s_prefix = excl;
    }
  }
  __syncthreads();
  uint32_t run = s_prefix + wbase + (incl - tsum);
#pragma unroll
  for (int e = 0; e < kScanItems; ++e) {
    if (base + e < n) out[base + e] = run;
    run += v[e];
  }
}

}  // namespace
