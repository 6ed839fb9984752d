// Latency price list of MI355X as the ESDF replay's rankings see it: one workgroup of 256 threads on an otherwise idle chip.
// hipcc --offload-arch=gfx950 -O3 -o lat_bench lat_bench.hip ; ./lat_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_chase(const uint32_t* next, uint32_t start, int steps, unsigned long long* out) {
  uint32_t i = start;
  const unsigned long long t0 = wall_clock64();
  for (int s = 0; s < steps; ++s) i = next[i];
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = i; }
}
__global__ void k_chase_atomic(uint32_t* next, uint32_t start, int steps, unsigned long long* out) {
  uint32_t i = start;
  const unsigned long long t0 = wall_clock64();
  for (int s = 0; s < steps; ++s) i = atomicAdd(&next[i], 0u);
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = i; }
}
__global__ void k_chase_exch(uint32_t* next, uint32_t* flags, uint32_t start, int steps, unsigned long long* out) {
  uint32_t i = start;
  const unsigned long long t0 = wall_clock64();
  for (int s = 0; s < steps; ++s) { const uint32_t w = atomicExch(&flags[i], 1u); i = next[i] + w; }   // load + atomic in one trip (independent)
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = i; }
}
__global__ void k_barrier(int steps, unsigned long long* out) {
  __shared__ uint32_t x;
  const unsigned long long t0 = wall_clock64();
  for (int s = 0; s < steps; ++s) { if (threadIdx.x == (s & 255)) x = s; __syncthreads(); }
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = x; }
}
__global__ void k_clock(int steps, unsigned long long* out) {
  unsigned long long acc = 0;
  const unsigned long long t0 = wall_clock64();
  for (int s = 0; s < steps; ++s) acc += wall_clock64();
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = acc; }
}
// store then barrier (the fence of __syncthreads waits for the store's acknowledgement)
__global__ void k_store_barrier(uint32_t* buf, int steps, unsigned long long* out) {
  const unsigned long long t0 = wall_clock64();
  for (int s = 0; s < steps; ++s) { buf[(size_t)s * 4096 + threadIdx.x * 16] = s; __syncthreads(); }
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; }
}
// every thread loads 1 / 4 / 8 independent random words per trip, then a barrier: what a "batched" pass costs
template <int B>
__global__ void k_batch(const uint32_t* next, int steps, unsigned long long* out) {
  uint32_t i[B];
  for (int b = 0; b < B; ++b) i[b] = (threadIdx.x * 977u + b * 131071u + blockIdx.x * 7919u) & 0xFFFFFu;
  const unsigned long long t0 = wall_clock64();
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int b = 0; b < B; ++b) i[b] = next[i[b]];
  }
  const unsigned long long t1 = wall_clock64();
  uint32_t acc = 0;
  for (int b = 0; b < B; ++b) acc += i[b];
  if (threadIdx.x == 0) { out[0] = t1 - t0; }
  if (acc == 0xdeadbeef) out[1] = acc;
}
__global__ void k_busy(float* x, int iters) {   // background load: other workgroups spinning on ALU + memory
  float v = x[blockIdx.x * blockDim.x + threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  x[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

// hot counters: every wave of a 1,024 x 256 grid takes `per` tickets (lane 0, returning atomicAdd, the ticket is used for a store) from
// one of `shards` counters `stride` words apart
__global__ void k_tickets(uint32_t* counters, uint32_t* sink, int shards, int stride, int per) {
  if ((threadIdx.x & 63) != 0) return;
  const uint32_t wv = blockIdx.x * 4 + (threadIdx.x >> 6);
  uint32_t* c = counters + (size_t)(wv % shards) * stride;
  for (int k = 0; k < per; ++k) {
    const uint32_t t = atomicAdd(c, 1u);
    sink[(wv * 64 + (t & 63)) & 0xFFFFF] = t;
  }
}

// the replay's step kernel reduced to its skeleton: every workgroup reads a header word; the first `active` ones do `trips` dependent
// loads, arrive at a counter, and the last one copies `words` words through LDS with atomic reads, lets thread 0 do some scalar work on
// them and stores them back
__global__ void __launch_bounds__(256) k_step(uint32_t* ctl, const uint32_t* next, uint32_t active, int trips, int words, uint32_t seq) {
  __shared__ uint32_t lds[1024];
  __shared__ uint32_t last;
  const uint32_t hdr = __hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  if (hdr != seq) return;
  if (blockIdx.x >= active) return;
  uint32_t i = (blockIdx.x * 256 + threadIdx.x) & 0xFFFFF;
  for (int t = 0; t < trips; ++t) i = next[i];
  if (i == 0xdeadbeef) ctl[2000] = i;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(&ctl[64], 1u) == active - 1 ? 1u : 0u;
  __syncthreads();
  if (!last) return;
  for (int w = threadIdx.x; w < words; w += 256) lds[w] = atomicAdd(&ctl[128 + w], 0u);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (int k = 0; k < 200; ++k) acc += lds[(k * 7) & 127];
    lds[1] = acc;
    lds[0] = seq + 1;
  }
  __syncthreads();
  for (int w = threadIdx.x; w < words; w += 256) ctl[128 + w] = lds[w];
  if (threadIdx.x == 0) { ctl[64] = 0; *reinterpret_cast<volatile uint32_t*>(&ctl[0]) = seq + 1; }
}

static std::vector<uint32_t> cycle(size_t n, uint32_t seed) {
  std::vector<uint32_t> perm(n), next(n);
  std::iota(perm.begin(), perm.end(), 0u);
  std::mt19937 g(seed);
  std::shuffle(perm.begin(), perm.end(), g);
  for (size_t k = 0; k < n; ++k) next[perm[k]] = perm[(k + 1) % n];
  return next;
}

int main() {
  unsigned long long* out; CK(hipMalloc(&out, 64));
  unsigned long long h[2];
  auto report = [&](const char* what, int steps) {
    CK(hipDeviceSynchronize()); CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
    printf("%-64s %8.1f ns per step\n", what, h[0] * 10.0 / steps);
  };
  const int steps = 2000;
  for (size_t words : {size_t(1) << 16, size_t(1) << 20, size_t(1) << 24, size_t(1) << 28}) {
    auto nx = cycle(words, 7);
    uint32_t *d, *fl; CK(hipMalloc(&d, words * 4)); CK(hipMalloc(&fl, words * 4));
    CK(hipMemcpy(d, nx.data(), words * 4, hipMemcpyHostToDevice)); CK(hipMemset(fl, 0, words * 4));
    char name[128];
    for (int rep = 0; rep < 2; ++rep) {
      snprintf(name, sizeof name, "dependent loads, 1 thread, %zu KB%s", words * 4 / 1024, rep ? " (again)" : "");
      k_chase<<<1, 1>>>(d, 0, steps, out); report(name, steps);
    }
    snprintf(name, sizeof name, "dependent loads, 256 threads same address, %zu KB", words * 4 / 1024);
    k_chase<<<1, 256>>>(d, 0, steps, out); report(name, steps);
    snprintf(name, sizeof name, "dependent returning atomicAdd(+0), 1 thread, %zu KB", words * 4 / 1024);
    k_chase_atomic<<<1, 1>>>(d, 0, steps, out); report(name, steps);
    snprintf(name, sizeof name, "load + atomicExch per step, 1 thread, %zu KB", words * 4 / 1024);
    k_chase_exch<<<1, 1>>>(d, fl, 0, steps, out); report(name, steps);
    if (words >= (size_t(1) << 20)) {
      snprintf(name, sizeof name, "256 threads x 1 random load per step, %zu KB", words * 4 / 1024);
      k_batch<1><<<1, 256>>>(d, steps, out); report(name, steps);
      snprintf(name, sizeof name, "256 threads x 4 random loads per step, %zu KB", words * 4 / 1024);
      k_batch<4><<<1, 256>>>(d, steps, out); report(name, steps);
      snprintf(name, sizeof name, "256 threads x 8 random loads per step, %zu KB", words * 4 / 1024);
      k_batch<8><<<1, 256>>>(d, steps, out); report(name, steps);
      snprintf(name, sizeof name, "... the same in 150 workgroups at once, x 8, %zu KB", words * 4 / 1024);
      k_batch<8><<<150, 256>>>(d, steps, out); report(name, steps);
    }
    CK(hipFree(d)); CK(hipFree(fl));
  }
  k_barrier<<<1, 256>>>(20000, out); report("__syncthreads, 256 threads", 20000);
  k_clock<<<1, 64>>>(20000, out); report("wall_clock64", 20000);
  {
    uint32_t* buf; CK(hipMalloc(&buf, size_t(2000) * 4096 * 4));
    k_store_barrier<<<1, 256>>>(buf, 2000, out); report("store + __syncthreads, 256 threads", 2000);
    CK(hipFree(buf));
  }
  {
    uint32_t *cn, *sink; CK(hipMalloc(&cn, 1 << 20)); CK(hipMalloc(&sink, (1 << 20) * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int per : {1, 4}) for (int stride : {1, 32, 1024}) for (int shards : {1, 4, 16, 64}) {
      if (shards == 1 && stride != 1) continue;
      CK(hipMemset(cn, 0, 1 << 20));
      k_tickets<<<1024, 256>>>(cn, sink, shards, stride, per);   // warm
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int rep = 0; rep < 20; ++rep) k_tickets<<<1024, 256>>>(cn, sink, shards, stride, per);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("tickets: 4096 waves x %d, %2d counters %4d words apart: %7.2f us per launch (%.0f tickets per us)\n", per, shards, stride, ms * 1000 / 20,
             4096.0 * per / (ms * 1000 / 20));
    }
  }
  {
    uint32_t* ctl; CK(hipMalloc(&ctl, 1 << 16));
    auto nx = cycle(size_t(1) << 20, 7);
    uint32_t* d; CK(hipMalloc(&d, nx.size() * 4)); CK(hipMemcpy(d, nx.data(), nx.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {1024, 512, 256, 64}) for (int active : {1, 64, 1024}) for (int trips : {0, 4}) for (int words : {0, 190, 725}) {
      if (active > grid) continue;
      if (words == 725 && !(active == 1 && trips == 0)) continue;
      CK(hipMemset(ctl, 0, 1 << 16));
      const int N = 2000;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(ctl, 0, 1 << 16));
        CK(hipEventRecord(e0));
        for (int k = 0; k < N; ++k) k_step<<<grid, 256>>>(ctl, d, active, trips, words, k);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      }
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      uint32_t h; CK(hipMemcpy(&h, ctl, 4, hipMemcpyDeviceToHost));
      printf("step kernel: grid %4d, %4d workgroups with work of %d trips, control moves %3d words: %6.2f us per launch%s\n", grid, active, trips, words, ms * 1000 / N, h == N ? "" : "  (header wrong!)");
    }
  }
  // with the rest of the chip busy
  {
    float* x; CK(hipMalloc(&x, 1023 * 256 * 4)); CK(hipMemset(x, 0, 1023 * 256 * 4));
    auto nx = cycle(size_t(1) << 24, 7);
    uint32_t* d; CK(hipMalloc(&d, nx.size() * 4)); CK(hipMemcpy(d, nx.data(), nx.size() * 4, hipMemcpyHostToDevice));
    hipStream_t s2; CK(hipStreamCreate(&s2));
    k_busy<<<1023, 256, 0, s2>>>(x, 4000000);
    k_chase<<<1, 1>>>(d, 0, steps, out); report("dependent loads, 1 thread, 64 MB, 1023 busy workgroups beside", steps);
    CK(hipDeviceSynchronize());
  }
  return 0;
}
