// vbx_sort.hpp — stable LSD radix sort of 64-bit keys (optionally with 32-bit values) on a bit field.
// Part of libvbx_hip.so's single translation unit (included by vbx_hip.hip, in order).
//
// Every sort of the hot path orders 3e5..1e6 keys by a 20..26-bit field and must be STABLE (the
// inputs are produced in visiting order).  rocPRIM's onesweep handles that size with 8-bit
// digits — 3-4 passes, each a launch plus two or three small memsets plus a histogram pass:
// ~105 us for a 20-bit field, all of it launch/latency.  Here a pass takes a digit of up to 12
// bits (20..24-bit fields in TWO passes) and consists of exactly three launches:
//   k_rsort_count    per-workgroup digit histogram (LDS atomics)      -> hist[digit][workgroup]
//   exclusive scan   over hist in digit-major order (rocPRIM scan)     -> global base per (digit, workgroup)
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
k_rsort_count(const uint64_t* __restrict__ keys, uint32_t n, int shift, uint32_t* __restrict__ hist, uint32_t nwg) {
  constexpr int NB = 1 << BITS;
  __shared__ uint32_t s_cnt[NB];
  for (int i = threadIdx.x; i < NB; i += kSortThreads) s_cnt[i] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * kSortTile;
#pragma unroll
  for (int e = 0; e < kSortItems; ++e) {
    const uint32_t i = base + e * kSortThreads + threadIdx.x;  // order is irrelevant for counting
    if (i < n) atomicAdd(&s_cnt[(uint32_t)(keys[i] >> shift) & (NB - 1)], 1u);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < NB; d += kSortThreads) hist[(size_t)d * nwg + blockIdx.x] = s_cnt[d];
}

template <int BITS, bool kHasVals>
__global__ void __launch_bounds__(kSortThreads)
k_rsort_scatter(const uint64_t* __restrict__ kin, const uint32_t* __restrict__ vin, uint64_t* __restrict__ kout,
                uint32_t* __restrict__ vout, uint32_t n, int shift, const uint32_t* __restrict__ gofs, uint32_t nwg) {
  constexpr int NB = 1 << BITS;
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
  for (int c = 0; c < kSortItems; ++c) {
    const uint32_t i = wbase + c * 64 + lane;
    key[c] = (i < n) ? kin[i] : 0;
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

}  // namespace
