// vbx_shard.cc — libvbx_shard.so: ray-bundle sharding over RCCL on top of the C-ABI of vbx_hip.h
// (include/vbx_shard.h has the protocol).  Host code only: the kernels it needs (export of the touched
// blocks' weighted sums, owner-side merge with duplicate rows) live in libvbx_hip.so.
#include "../../include/vbx_shard.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

struct vbx_shard {
  vbx_ctx* p = nullptr;
  vbx_ctx* d = nullptr;                 // delta 0 (the one given to vbx_shard_create)
  std::vector<vbx_ctx*> deltas;         // every registered delta map, d first
  int n_sets = 1;                       // 2: the deltas are two alternating sets and the exchange runs behind the next step
  int cur_set = 0;                      // the set the current step integrates into
  size_t used_in_set = 1;               // deltas of the current set that received shards this step
  std::thread worker;                   // the previous step's exchange (pipelined mode)
  int worker_rc = VBX_OK;
  bool worker_running = false;
  int rank = 0, world = 1, device = 0;
  ncclComm_t comm = nullptr;
  bool dead = false;  // the communicator was aborted (world > 1): no step may run any more
  hipStream_t stream = nullptr;
  size_t nvox = 4096;
  float* d_send = nullptr;
  float* d_recv = nullptr;
  size_t send_cap = 0, recv_cap = 0;  // rows
  int32_t* d_keys_send = nullptr;
  int32_t* d_keys_recv = nullptr;
  size_t ks_cap = 0, kr_cap = 0;      // rows
  unsigned long long* d_counts = nullptr;  // world rows of [counts to each rank..., status]
  vbx_shard_stats stats{};
  std::string err;
  void fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    err = buf;
  }
};

namespace {
thread_local std::string g_err;

#define HIPS(expr)                                                                  \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess) {                                                         \
      s->fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return VBX_ERR_HIP;                                                           \
    }                                                                               \
  } while (0)
#define NCCLS(expr)                                                                 \
  do {                                                                              \
    ncclResult_t _e = (expr);                                                       \
    if (_e != ncclSuccess) {                                                        \
      s->fail("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(_e), __FILE__, __LINE__); \
      return VBX_ERR_HIP;                                                           \
    }                                                                               \
  } while (0)
#define VBXS(ctx, expr)                                       \
  do {                                                        \
    int _rc = (expr);                                         \
    if (_rc != VBX_OK) {                                      \
      s->fail("%s: %s", #expr, vbx_last_error(ctx));          \
      return _rc;                                             \
    }                                                         \
  } while (0)

template <typename T>
int grow(vbx_shard* s, T** p, size_t* cap, size_t want, size_t elems_per_row) {
  if (want <= *cap) return VBX_OK;
  if (*p) HIPS(hipFree(*p));
  *p = nullptr;
  const size_t rows = std::max(want, *cap + *cap / 2 + 16);
  HIPS(hipMalloc((void**)p, rows * elems_per_row * sizeof(T)));
  *cap = rows;
  return VBX_OK;
}
}  // namespace

// a row = the delta block itself: distance, weight and colour planes (vbx_blocks_export_sums); 48 KiB at vps 16
static constexpr size_t kRowPlanes = 3;

extern "C" {

int vbx_shard_owner_of(const int32_t idx[3], int world) {
  const long long h = ((long long)idx[0] * 73856093ll) ^ ((long long)idx[1] * 19349663ll) ^ ((long long)idx[2] * 83492791ll);
  return (int)((h & 0x7FFFFFFFll) % (long long)std::max(world, 1));
}

int vbx_shard_get_unique_id(uint8_t id[VBX_SHARD_ID_BYTES]) {
  static_assert(VBX_SHARD_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  ncclUniqueId u;
  if (ncclGetUniqueId(&u) != ncclSuccess) return VBX_ERR_HIP;
  std::memcpy(id, u.internal, VBX_SHARD_ID_BYTES);
  return VBX_OK;
}

const char* vbx_shard_last_error(vbx_shard* s) { return s ? s->err.c_str() : g_err.c_str(); }

vbx_shard* vbx_shard_create(vbx_ctx* persistent, vbx_ctx* delta, int rank, int world, const uint8_t id[VBX_SHARD_ID_BYTES],
                            int device) {
  if (!persistent || !delta || world < 1 || rank < 0 || rank >= world || (world > 1 && !id)) {
    g_err = "vbx_shard_create: bad argument";
    return nullptr;
  }
  vbx_shard* s = new vbx_shard;
  s->p = persistent;
  s->d = delta;
  s->deltas.push_back(delta);
  s->rank = rank;
  s->world = world;
  s->device = device;
  bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) == hipSuccess;
  if (ok && id) {  // also with world == 1 when an id is given: the collectives then run for real on one rank
    ncclUniqueId u;
    std::memcpy(u.internal, id, VBX_SHARD_ID_BYTES);
    ok = ncclCommInitRank(&s->comm, world, u, rank) == ncclSuccess &&
         hipMalloc((void**)&s->d_counts, (size_t)world * (world + 1) * sizeof(unsigned long long)) == hipSuccess;
  }
  if (!ok) {
    g_err = "vbx_shard_create: stream / RCCL communicator initialisation failed";
    vbx_shard_destroy(s);
    return nullptr;
  }
  vbx_map_cfg cp{}, cd{};
  if (vbx_get_map_cfg(persistent, &cp) != VBX_OK || vbx_get_map_cfg(delta, &cd) != VBX_OK ||
      cp.voxels_per_side != cd.voxels_per_side || cp.voxel_size != cd.voxel_size) {
    g_err = "vbx_shard_create: persistent and delta map must have the same voxel size and voxels per side";
    vbx_shard_destroy(s);
    return nullptr;
  }
  s->nvox = (size_t)cp.voxels_per_side * cp.voxels_per_side * cp.voxels_per_side;
  (void)vbx_set_block_order_tracking(delta, 0);   // a delta map is rebuilt every step; nobody asks for its Layer order
  return s;
}

void vbx_shard_destroy(vbx_shard* s) {
  if (!s) return;
  if (s->worker.joinable()) s->worker.join();
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  if (s->comm) (void)ncclCommDestroy(s->comm);
  for (void* q : {(void*)s->d_send, (void*)s->d_recv, (void*)s->d_keys_send, (void*)s->d_keys_recv, (void*)s->d_counts})
    if (q) (void)hipFree(q);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

static size_t set_size(const vbx_shard* s) { return s->deltas.size() / (size_t)s->n_sets; }
static vbx_ctx* delta_of(const vbx_shard* s, int set, size_t u) { return s->deltas[(size_t)set * set_size(s) + u]; }

static int join_worker(vbx_shard* s) {
  if (s->worker.joinable()) s->worker.join();
  s->worker_running = false;
  const int rc = s->worker_rc;
  s->worker_rc = VBX_OK;
  return rc;
}

int vbx_shard_add_delta(vbx_shard* s, vbx_ctx* delta) {
  if (!s || !delta) return VBX_ERR_INVALID;
  vbx_map_cfg c0{}, c1{};
  if (vbx_get_map_cfg(s->d, &c0) != VBX_OK || vbx_get_map_cfg(delta, &c1) != VBX_OK || c0.voxels_per_side != c1.voxels_per_side ||
      c0.voxel_size != c1.voxel_size) {
    s->fail("vbx_shard_add_delta: every delta map must have the persistent map's voxel size and voxels per side");
    return VBX_ERR_INVALID;
  }
  s->deltas.push_back(delta);
  (void)vbx_set_block_order_tracking(delta, 0);
  return VBX_OK;
}

int vbx_shard_set_pipelined(vbx_shard* s, int on) {
  if (!s) return VBX_ERR_INVALID;
  int rc = join_worker(s);
  if (rc) return rc;
  if (on && (s->deltas.size() < 2 || s->deltas.size() % 2)) {
    s->fail("vbx_shard_set_pipelined: needs an even number of delta maps (two alternating sets)");
    return VBX_ERR_INVALID;
  }
  s->n_sets = on ? 2 : 1;
  s->cur_set = 0;
  return VBX_OK;
}

int vbx_shard_wait(vbx_shard* s) {
  if (!s) return VBX_ERR_INVALID;
  return join_worker(s);
}

int vbx_shard_begin_step(vbx_shard* s) {
  if (!s) return VBX_ERR_INVALID;
  if (s->dead) { s->err = "the shard's communicator was aborted after an allocation failure: no further step can run"; return VBX_ERR_HIP; }
  // pipelined: the exchange of the previous step may still read the OTHER set; the one that used THIS set (two steps
  // ago) was joined by the end_step in between
  for (size_t u = 0; u < set_size(s); ++u) VBXS(delta_of(s, s->cur_set, u), vbx_clear_keep_slots(delta_of(s, s->cur_set, u)));
  s->used_in_set = 1;
  return VBX_OK;
}

int vbx_shard_integrate(vbx_shard* s, int kind, const vbx_tsdf_cfg* cfg, const float pos[3], const float quat[4],
                        const float* d_points_C, const uint8_t* d_rgba, size_t n, int freespace_points) {
  if (!s) return VBX_ERR_INVALID;
  vbx_ctx* d = delta_of(s, s->cur_set, 0);
  VBXS(d, vbx_tsdf_integrate_device(d, kind, cfg, pos, quat, d_points_C, d_rgba, n, freespace_points));
  return VBX_OK;
}

int vbx_shard_integrate_shards(vbx_shard* s, int kind, const vbx_tsdf_cfg* cfg, size_t n_shards, const float* pos_xyz,
                               const float* quat_wxyz, const float* const* d_points_C, const uint8_t* const* d_rgba,
                               const size_t* n_points, int freespace_points) {
  if (!s || !cfg || (n_shards && (!pos_xyz || !quat_wxyz || !d_points_C || !d_rgba || !n_points))) return VBX_ERR_INVALID;
  const size_t nd = set_size(s);
  const size_t used = std::max<size_t>(1, std::min(nd, n_shards));
  s->used_in_set = std::max(s->used_in_set, used);
  std::vector<int> rcs(used, VBX_OK);
  auto run = [&](size_t u) {  // shard i goes into delta i % nd; the shards of one delta one after the other
    (void)hipSetDevice(s->device);
    vbx_ctx* d = delta_of(s, s->cur_set, u);
    for (size_t i = u; i < n_shards && rcs[u] == VBX_OK; i += nd)
      rcs[u] = vbx_tsdf_integrate_device(d, kind, cfg, pos_xyz + 3 * i, quat_wxyz + 4 * i, d_points_C[i], d_rgba[i], n_points[i],
                                         freespace_points);
  };
  std::vector<std::thread> th;
  for (size_t u = 1; u < used; ++u) th.emplace_back(run, u);  // one host thread + one HIP stream per delta map
  run(0);
  for (std::thread& t : th) t.join();
  for (size_t u = 0; u < used; ++u)
    if (rcs[u] != VBX_OK) {
      s->fail("vbx_shard_integrate_shards: delta %zu: %s", u, vbx_last_error(delta_of(s, s->cur_set, u)));
      return rcs[u];
    }
  return VBX_OK;
}

// Everything a rank does BEFORE the first collective of a step: list the delta's blocks grouped by owner and export
// their weighted sums.  A failure here must not leave the other ranks blocked in the collectives, so the status
// travels with the group sizes (vbx_shard_end_step).
static int prepare_step(vbx_shard* s, int set, size_t used, std::vector<int32_t>& send_keys, std::vector<size_t>& send_counts,
                        std::vector<size_t>& sdispl, size_t* n_out) {
  // 1. the blocks this step's deltas touched: rows owner-major, then delta order, then (z,y,x) — at the owner the rows
  //    of one block arrive in (sender rank, delta, key) order = the global shard order, whatever the world size
  const int W = s->world;
  std::vector<std::vector<int32_t>> idx(used);
  std::vector<std::vector<int>> owner(used);
  std::vector<std::vector<size_t>> cnt(used, std::vector<size_t>(W, 0));
  send_counts.assign(W, 0);
  for (size_t u = 0; u < used; ++u) {
    vbx_ctx* d = delta_of(s, set, u);
    size_t n = 0;
    VBXS(d, vbx_num_blocks(d, VBX_LAYER_TSDF, &n));
    idx[u].assign(3 * std::max<size_t>(n, 1), 0);
    if (n) VBXS(d, vbx_block_indices(d, VBX_LAYER_TSDF, idx[u].data(), n, &n));  // ascending (z,y,x)
    idx[u].resize(3 * n);
    owner[u].resize(n);
    for (size_t i = 0; i < n; ++i) {
      owner[u][i] = vbx_shard_owner_of(&idx[u][3 * i], W);
      ++cnt[u][owner[u][i]];
      ++send_counts[owner[u][i]];
    }
  }
  sdispl.assign(W + 1, 0);
  for (int r = 0; r < W; ++r) sdispl[r + 1] = sdispl[r] + send_counts[r];
  const size_t n_total = sdispl[W];
  send_keys.assign(3 * std::max<size_t>(n_total, 1), 0);
  // start row of segment (owner o, delta u)
  std::vector<std::vector<size_t>> seg(used, std::vector<size_t>(W, 0));
  for (int o = 0; o < W; ++o) {
    size_t at = sdispl[o];
    for (size_t u = 0; u < used; ++u) {
      seg[u][o] = at;
      at += cnt[u][o];
    }
  }
  for (size_t u = 0; u < used; ++u) {
    std::vector<size_t> cur(seg[u]);
    for (size_t i = 0; i < owner[u].size(); ++i) {  // stable: the (z,y,x) order survives inside every segment
      const size_t at = cur[owner[u][i]]++;
      std::memcpy(&send_keys[3 * at], &idx[u][3 * i], 12);
    }
  }
  // 2. their weighted sums, in the same order: one export per (delta, owner) segment (one per delta when there is one)
  int rc = grow(s, &s->d_send, &s->send_cap, n_total, kRowPlanes * s->nvox);
  if (rc) return rc;
  for (size_t u = 0; u < used; ++u) {
    vbx_ctx* d = delta_of(s, set, u);
    if (used == 1) {
      if (n_total) VBXS(d, vbx_blocks_export_sums(d, send_keys.data(), n_total, s->d_send));
      break;
    }
    for (int o = 0; o < W; ++o)
      if (cnt[u][o])
        VBXS(d, vbx_blocks_export_sums(d, &send_keys[3 * seg[u][o]], cnt[u][o], s->d_send + seg[u][o] * kRowPlanes * s->nvox));
  }
  *n_out = n_total;
  return VBX_OK;
}

static int exchange_and_merge(vbx_shard* s, int set, size_t used, int apply_caps, float truncation_distance, float max_weight);

int vbx_shard_end_step(vbx_shard* s, int apply_caps, float truncation_distance, float max_weight) {
  if (!s) return VBX_ERR_INVALID;
  if (s->dead) { s->err = "the shard's communicator was aborted after an allocation failure: no further step can run"; return VBX_ERR_HIP; }
  // the exchange of the step before this one (pipelined mode) must be through: collectives are issued in step order
  int rc = join_worker(s);
  if (rc) return rc;
  const int set = s->cur_set;
  const size_t used = s->used_in_set;
  if (s->n_sets == 1) return exchange_and_merge(s, set, used, apply_caps, truncation_distance, max_weight);
  // pipelined: this step's exchange runs on a worker thread (its own stream) while the caller integrates the next
  // step into the other set of delta maps; its status surfaces at the next end_step / vbx_shard_wait
  s->cur_set ^= 1;
  s->worker_running = true;
  s->worker = std::thread([=]() { s->worker_rc = exchange_and_merge(s, set, used, apply_caps, truncation_distance, max_weight); });
  return VBX_OK;
}

static int exchange_and_merge(vbx_shard* s, int set, size_t used, int apply_caps, float truncation_distance, float max_weight) {
  HIPS(hipSetDevice(s->device));
  const int W = s->world;
  std::vector<int32_t> send_keys;
  std::vector<size_t> send_counts, sdispl;
  size_t n = 0;
  const int local_rc = prepare_step(s, set, used, send_keys, send_counts, sdispl, &n);
  if (!s->comm) {
    if (local_rc) return local_rc;
  }
  const size_t nvox = s->nvox;
  const float* d_in = s->d_send;
  std::vector<int32_t> recv_keys;
  size_t n_recv = n;
  if (!s->comm) {
    recv_keys = send_keys;
  } else {
    // 3. group sizes: all-gather of every rank's row [send_counts..., status] -> counts[sender][dest], status[sender].
    //    A rank that failed locally sends zero counts and its error code: every rank sees it in the same
    //    collective and all of them leave the step together (no rank is left waiting in the all-to-all).
    const size_t RW = (size_t)W + 1;
    std::vector<unsigned long long> row(RW, 0ull);
    if (local_rc == VBX_OK)
      for (int r = 0; r < W; ++r) row[r] = send_counts[r];
    row[W] = (unsigned long long)(long long)local_rc;
    HIPS(hipMemcpyAsync(s->d_counts + (size_t)s->rank * RW, row.data(), RW * sizeof(unsigned long long), hipMemcpyHostToDevice, s->stream));
    NCCLS(ncclAllGather(s->d_counts + (size_t)s->rank * RW, s->d_counts, RW, ncclUint64, s->comm, s->stream));
    std::vector<unsigned long long> all((size_t)W * RW);
    HIPS(hipMemcpyAsync(all.data(), s->d_counts, all.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream));
    HIPS(hipStreamSynchronize(s->stream));
    for (int r = 0; r < W; ++r) {
      const int st = (int)(long long)all[(size_t)r * RW + W];
      if (st != VBX_OK) {
        if (r != s->rank) s->fail("vbx_shard_end_step: rank %d failed before the exchange (status %d); step abandoned on every rank", r, st);
        return local_rc ? local_rc : st;
      }
    }
    std::vector<size_t> recv_counts(W), rdispl(W + 1, 0);
    for (int r = 0; r < W; ++r) {
      recv_counts[r] = (size_t)all[(size_t)r * RW + s->rank];
      rdispl[r + 1] = rdispl[r] + recv_counts[r];
    }
    n_recv = rdispl[W];
    // 4. BlockIndex rows, then the sums: sparse all-to-all-v (rows arrive grouped by sender rank).  The only local
    //    failure left between here and the collectives is device memory for the receive side: that rank aborts the
    //    communicator, so that the others' collectives return an error instead of waiting for it.
    int rc = grow(s, &s->d_keys_send, &s->ks_cap, n, 3);
    if (!rc) rc = grow(s, &s->d_keys_recv, &s->kr_cap, n_recv, 3);
    if (!rc) rc = grow(s, &s->d_recv, &s->recv_cap, n_recv, kRowPlanes * nvox);
    if (rc) {
      (void)ncclCommAbort(s->comm);
      s->comm = nullptr;
      s->dead = true;   // every later step of this shard fails: without its communicator it must not merge other owners' blocks
      return rc;
    }
    if (n) HIPS(hipMemcpyAsync(s->d_keys_send, send_keys.data(), n * 12, hipMemcpyHostToDevice, s->stream));
    std::vector<size_t> sc(W), sd(W), rcn(W), rd(W);
    for (int r = 0; r < W; ++r) { sc[r] = send_counts[r] * 3; sd[r] = sdispl[r] * 3; rcn[r] = recv_counts[r] * 3; rd[r] = rdispl[r] * 3; }
    NCCLS(ncclAllToAllv(s->d_keys_send, sc.data(), sd.data(), s->d_keys_recv, rcn.data(), rd.data(), ncclInt32, s->comm, s->stream));
    const size_t row_f = kRowPlanes * nvox;
    for (int r = 0; r < W; ++r) { sc[r] = send_counts[r] * row_f; sd[r] = sdispl[r] * row_f; rcn[r] = recv_counts[r] * row_f; rd[r] = rdispl[r] * row_f; }
    NCCLS(ncclAllToAllv(s->d_send, sc.data(), sd.data(), s->d_recv, rcn.data(), rd.data(), ncclFloat32, s->comm, s->stream));
    recv_keys.resize(3 * std::max<size_t>(n_recv, 1));
    if (n_recv) HIPS(hipMemcpyAsync(recv_keys.data(), s->d_keys_recv, n_recv * 12, hipMemcpyDeviceToHost, s->stream));
    HIPS(hipStreamSynchronize(s->stream));
    d_in = s->d_recv;
  }
  // 5. owner merge: rows of one block are added in (sender rank, key) order, then merged once
  if (n_recv) VBXS(s->p, vbx_blocks_merge_sums(s->p, recv_keys.data(), n_recv, d_in, apply_caps, truncation_distance, max_weight));
  ++s->stats.steps;
  s->stats.sent_blocks += n;
  s->stats.received_blocks += n_recv;
  s->stats.payload_bytes += (uint64_t)n * kRowPlanes * nvox * 4;
  return VBX_OK;
}

int vbx_shard_get_stats(vbx_shard* s, vbx_shard_stats* out) {
  if (!s || !out) return VBX_ERR_INVALID;
  *out = s->stats;
  return VBX_OK;
}

}  // extern "C"
