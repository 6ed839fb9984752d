// vbx_host_tsdf.hpp — host orchestration of the three TSDF integrators (kernel launch sequences, host decisions)
// Part of libvbx_hip.so's single translation unit (included by vbx_hip.hip, in order).

namespace {
// Visiting order of the points: nullptr = MixedThreadSafeIndex (closed form on the device), else
// a device array s_of_p for "sorted" (integration_order_mode, tsdf_integrator.h:72-74).
int visiting_order(vbx_ctx* ctx, const vbx_tsdf_cfg* cfg, const float* d_pts, size_t n, const uint32_t** order) {
  *order = nullptr;
  if (cfg->integration_order_mode == 0) return VBX_OK;
  HIP_TRY(ctx->b_keys0.ensure(n * 8));
  HIP_TRY(ctx->b_keys1.ensure(n * 8));
  HIP_TRY(ctx->b_order.ensure(n * 4));
  KLAUNCH(k_sorted_keys, grid_for(n), dim3(256), 0, ctx->stream, d_pts, n, ctx->b_keys0.as<uint64_t>());
  int rc = stable_sort01(ctx, n, 0, 64, false);
  if (rc) return rc;
  KLAUNCH(k_sorted_inverse, grid_for(n), dim3(256), 0, ctx->stream, ctx->b_keys1.as<uint64_t>(), n,
                     ctx->b_order.as<uint32_t>());
  *order = ctx->b_order.as<uint32_t>();
  return VBX_OK;
}

// Order the emitted (voxel, order) keys and fold them per voxel (keys in b_keys0).
int sort_and_fold(vbx_ctx* ctx, const RayTab& tab, const CastCfg& c, uint32_t total, bool giant_runs = true) {
  MapDev& m = ctx->map;
  hipStream_t s = ctx->stream;
  // gids are below pool_used * nvox; h_state holds the value from after this call's slot
  // assignment (every path reads DevState back between k_commit_alloc and here)
  const unsigned end_bit = 32 + bits_for((uint64_t)std::max<uint32_t>(ctx->h_state.pool_used, 1) * m.nvox);
  // keys are emitted ray by ray in visiting order, so a stable sort on the voxel field alone
  // leaves every voxel's updates in visiting order (invalid keys, all ones, go last)
  int rc = stable_sort01(ctx, total, 32, std::min(64u, end_bit + 1), false);
  if (rc) return rc;
  tmark(ctx, 5);
  if (!giant_runs) {  // Fast: two or three updates per voxel (measured: 0.3-0.9 M updates on 0.1-0.4 M voxels)
    KLAUNCH(k_fold_direct, grid_for(total), dim3(256), 0, s, ctx->b_keys1.as<uint64_t>(), (size_t)total, tab, c, m,
                       ctx->d_state);
    tmark(ctx, 6);
    ctx->counters.voxel_updates = total;
    return VBX_OK;
  }
  // long runs are collected by k_fold and folded wave-cooperatively afterwards (their number is
  // bounded by total / kFoldShort)
  const uint32_t long_cap = total / kFoldShort + 2;  // per stripe (any stripe could hold all of them)
  // giant runs (one update per ray on the voxels around the sensor: Simple and Merged; Fast updates a voxel
  // once per call unless its approximate set forgets it) go to a workgroup each
  giant_runs = total > kFoldGiant;
  HIP_TRY(ctx->b_long.ensure(((size_t)long_cap * 16 + kGiantCap) * 4 + (size_t)total / 256 + 1));
  uint32_t* giant_list = giant_runs ? ctx->b_long.as<uint32_t>() + (size_t)long_cap * 16 : nullptr;
  uint8_t* ident = giant_runs ? reinterpret_cast<uint8_t*>(ctx->b_long.as<uint32_t>() + (size_t)long_cap * 16 + kGiantCap) : nullptr;
  HIP_TRY(ctx->b_fin.ensure((size_t)total * 12));
  float* in_sdf = ctx->b_fin.as<float>();
  float* in_uw = in_sdf + total;
  uint32_t* in_col = reinterpret_cast<uint32_t*>(in_uw + total);
  KLAUNCH(k_fold_inputs, grid_for(total), dim3(256), 0, s, ctx->b_keys1.as<uint64_t>(), (size_t)total, tab, c,
                     m, in_sdf, in_uw, in_col, ident);
  KLAUNCH(k_fold<16>, grid_for(total, 256 * 16), dim3(256), 0, s, ctx->b_keys1.as<uint64_t>(), (size_t)total, c, m,
                     in_sdf, in_uw, in_col, ctx->b_long.as<uint32_t>(), long_cap, giant_list, ctx->d_state);
  {
    const unsigned giant_blocks = giant_runs ? 128 : 0;
    const unsigned waves = (unsigned)std::min<size_t>(8192, (size_t)total / kFoldShort + 1);
    KLAUNCH(k_fold_runs, dim3(giant_blocks + (waves + kGiantWaves - 1) / kGiantWaves), dim3(64 * kGiantWaves), 0, s,
                       ctx->b_keys1.as<uint64_t>(), (size_t)total, c, m, in_sdf, in_uw, in_col, ctx->b_long.as<uint32_t>(),
                       long_cap, giant_list, ident, giant_blocks, ctx->d_state);
  }
  tmark(ctx, 6);
  ctx->counters.voxel_updates = total;
  return VBX_OK;
}

// Shared tail of all three integrators: allocate blocks along the rays, emit ordered voxel
// keys, sort, fold.  `limit` (optional) bounds the number of voxels each ray visits.
int march_and_fold(vbx_ctx* ctx, const RayTab& tab, const CastCfg& c, bool from_origin,
                   const uint32_t* limit, bool blocks_already_marked, const uint64_t* graze_keys,
                   uint32_t n_graze, bool two_pass = false) {
  const uint32_t R = tab.R;
  if (R == 0) return VBX_OK;
  MapDev& m = ctx->map;
  hipStream_t s = ctx->stream;

  HIP_TRY(ctx->b_cnt.ensure((size_t)(R + 1) * 4));
  HIP_TRY(ctx->b_off.ensure((size_t)(R + 1) * 4));
  // 32-bit key offsets: rays x (longest possible path) below 2^32 cannot wrap; only beyond that is the
  // exact 64-bit total worth its atomics
  const double seg_len = (double)c.max_ray_length_m + (double)c.trunc;
  const double per_ray_bound = 1.7320508075688772 * seg_len * (double)m.voxel_size_inv + 6.0;
  const bool may_wrap = (double)R * per_ray_bound > 4.0e9;
  KLAUNCH(k_ray_count, grid_for(R + 1), dim3(256), 0, s, tab, c, m, from_origin ? 1 : 0,
                     limit, ctx->b_cnt.as<uint32_t>(), 1 | (may_wrap ? 2 : 0), ctx->d_state);
  int rc = exclusive_scan_u32(ctx, ctx->b_cnt.as<uint32_t>(), ctx->b_off.as<uint32_t>(), R + 1);
  if (rc) return rc;
  uint32_t total = 0;
  for (;;) {
    if (!blocks_already_marked) {
      KLAUNCH(k_ray_mark_blocks, grid_for(R), dim3(256), 0, s, tab, c, m,
                         from_origin ? 1 : 0, limit, ctx->b_newlist.as<uint32_t>(), ctx->d_state);
      KLAUNCH(k_assign_slots, grid_for(m.cap_blocks), dim3(256), 0, s, m,
                         ctx->b_newlist.as<uint32_t>(), ctx->d_state);
      KLAUNCH(k_commit_alloc, dim3(1), dim3(1), 0, s, m, ctx->d_state);
      tmark(ctx, 2);
    }
    // total number of keys = off[R]
    rc = sync_state(ctx, ctx->b_off.as<uint32_t>() + R, &total);
    if (rc) return rc;
    if (!(ctx->h_state.error & 1u) || blocks_already_marked) break;
    rc = grow_pool(ctx);  // out of slots: double the pool and mark the blocks again
    if (rc) return rc;
  }
  if (may_wrap && ctx->h_state.total_keys > 0xFFFFFFF0ull) {  // the 32-bit offsets wrapped: no voxel has been written yet
    ctx->fail("cloud visits %llu voxels, more than one call can order (2^32): split the cloud",
              (unsigned long long)ctx->h_state.total_keys);
    return VBX_ERR_CAPACITY;
  }
  rc = check_state_error(ctx);
  if (rc) return rc;
  if (total == 0) return VBX_OK;

  HIP_TRY(ctx->b_keys0.ensure((size_t)total * 8));
  HIP_TRY(ctx->b_keys1.ensure((size_t)total * 8));
  KLAUNCH(k_ray_emit, grid_for(R), dim3(256), 0, s, tab, c, m, from_origin ? 1 : 0, limit,
                     ctx->b_off.as<uint32_t>(), ctx->b_keys0.as<uint64_t>(), graze_keys, n_graze, two_pass ? 1 : 0,
                     ctx->d_state);
  tmark(ctx, 4);
  return sort_and_fold(ctx, tab, c, total);
}

int integrate_simple(vbx_ctx* ctx, const vbx_tsdf_cfg* cfg, const Pose& T, const float* d_pts,
                     const uint32_t* d_rgba, size_t n, int freespace) {
  CastCfg c = make_cast_cfg(ctx, cfg, &T.t.x);
  const uint32_t* order = nullptr;
  {
    int orc_ = visiting_order(ctx, cfg, d_pts, n, &order);
    if (orc_) return orc_;
  }
  int rc = ensure_tab(ctx, false, n, false);
  if (rc) return rc;
  RayTab tab = make_tab(ctx, false, (uint32_t)n);
  tab.bkey = nullptr;
  KLAUNCH(k_prep_points, grid_for(n), dim3(256), 0, ctx->stream, d_pts, d_rgba, n, T, c,
                     freespace, tab, (float*)nullptr, (float*)nullptr, (float*)nullptr, order, ctx->map.voxel_size_inv,
                     (uint64_t*)nullptr, (uint32_t*)nullptr, (int32_t*)nullptr, ctx->d_state);
  tmark(ctx, 1);
  return march_and_fold(ctx, tab, c, /*from_origin=*/true, nullptr, false, nullptr, 0);
}

// MergedTsdfIntegrator::integrateVoxels walks voxel_map / clear_map — std::unordered_map keyed by
// GlobalIndex with LongIndexHash (block_hash.h:54-64) — from begin() to end()
// (tsdf_integrator.cc:440-456).  That order is a property of libstdc++'s hashtable (bucket
// count growth, node splicing) given the hash values and the insertion sequence — bundleRays' insertion order:
// first point of each bundle, visiting order — and is reconstructed on the host.
// perm[rank in ascending key order] = row in visiting order; non-clearing bundles first (:324-333).
//
// Inserting into the real container costs ~20 ns per bundle in node allocation and cache misses
// (0.6 ms per 640x480 frame).  The order itself follows from two facts of libstdc++'s _Hashtable
// (hashtable.h: _M_insert_bucket_begin and _M_rehash_aux for unique keys place a node the same
// way): a node whose bucket is empty goes to the head of the whole list, a node whose bucket is
// occupied goes to the head of its bucket's run.  So, for a fixed bucket count, the list is the
// buckets' runs in DESCENDING order of the bucket's first insertion, each run in DESCENDING
// insertion order; and a rehash re-inserts the current list, in list order, into the new bucket
// count before the insertion that triggered it.  The bucket counts and the element counts at
// which they change depend only on the number of elements (_Prime_rehash_policy), and are read
// off a real std::unordered_map once.  unordered_iteration_order() replays that with flat arrays;
// vbx_selftest_unordered_order checks it against the container itself.
struct HostU32Hash {
  size_t operator()(const std::pair<uint32_t, uint32_t>& k) const { return (size_t)k.first; }
};
// (number of elements already in the map, bucket count from that insertion on)
static std::vector<std::pair<uint32_t, uint32_t>> rehash_schedule(uint32_t n_needed) {
  static std::mutex mu;  // handles of different threads share the probe
  static std::unordered_map<uint32_t, char> probe;  // identity-hashed keys: the policy only counts
  static std::vector<std::pair<uint32_t, uint32_t>> pts;
  static uint32_t known = 0;
  std::lock_guard<std::mutex> lock(mu);
  while (known < n_needed) {
    const size_t before = probe.bucket_count();
    probe.emplace(known, 0);
    if (probe.bucket_count() != before) pts.emplace_back(known, (uint32_t)probe.bucket_count());
    ++known;
  }
  return pts;
}
// hashes[i] = hash of the i-th inserted (distinct) key; out[r] = insertion index of the r-th
// element in iteration order
struct OrderScratch {
  std::vector<uint32_t>*seq, *runs;
  std::vector<int32_t>*head, *nxt;
};
static void unordered_iteration_order(const uint32_t* hashes, uint32_t n, std::vector<uint32_t>* out,
                                      const OrderScratch& sc) {
  out->clear();
  if (n == 0) return;
  const auto sched = rehash_schedule(n);
  std::vector<uint32_t>& list = *out;  // current iteration order (insertion indices)
  std::vector<uint32_t>& seq = *sc.seq;
  std::vector<uint32_t>& run_order = *sc.runs;
  std::vector<int32_t>& head = *sc.head;
  std::vector<int32_t>& nxt = *sc.nxt;
  seq.reserve(n);
  list.reserve(n);
  uint32_t done = 0;  // elements inserted so far
  for (size_t k = 0; k < sched.size() && done < n; ++k) {
    if (sched[k].first >= n) break;
    const uint32_t B = sched[k].second;
    const uint32_t until = (k + 1 < sched.size()) ? std::min(sched[k + 1].first, n) : n;  // B holds for [done, until)
    // the phase's insertion sequence: the current list (rehash), then the new elements
    seq.assign(list.begin(), list.end());
    for (uint32_t i = done; i < until; ++i) seq.push_back(i);
    const uint32_t m = (uint32_t)seq.size();
    // counting sort by (run of the bucket, descending time) instead of chasing per-bucket chains:
    // pass 1 numbers the buckets in order of first use and sizes their runs, pass 2 (backwards in
    // time) drops every element at the next free place of its run; runs are laid out last-first
    head.assign(B, -1);            // bucket -> run number
    run_order.clear();             // run sizes, then run starts
    if (nxt.size() < m) nxt.resize(m);  // bucket of seq[t]
    const uint64_t M = ~0ull / B + 1;  // hash % B without a division (Lemire's fastmod, 32-bit operands)
    const uint32_t* sq = seq.data();
    for (uint32_t t = 0; t < m; ++t) {
      const uint32_t b = (uint32_t)(((unsigned __int128)(M * hashes[sq[t]]) * B) >> 64);
      int32_t r = head[b];
      if (r < 0) {
        r = (int32_t)run_order.size();
        head[b] = r;
        run_order.push_back(0);
      }
      ++run_order[(size_t)r];
      nxt[t] = r;
    }
    uint32_t acc = 0;
    for (size_t r = run_order.size(); r-- > 0;) {
      const uint32_t c = run_order[r];
      run_order[r] = acc;
      acc += c;
    }
    list.resize(m);
    for (uint32_t t = m; t-- > 0;) list[run_order[(size_t)nxt[t]]++] = sq[t];
    done = until;
  }
}
// the same from the container itself (self-test and the definition of "right")
static void unordered_iteration_order_real(const uint32_t* hashes, uint32_t n, std::vector<uint32_t>* out) {
  std::unordered_map<std::pair<uint32_t, uint32_t>, uint32_t, HostU32Hash> map;
  for (uint32_t i = 0; i < n; ++i) map.emplace(std::make_pair(hashes[i], i), i);
  out->clear();
  for (const auto& kv : map) out->push_back(kv.second);
}

// Two halves: _begin queues the device part and the copy of the insertion-order list to the host; _finish waits
// for that copy, replays the hashtable and uploads the permutation.  The caller queues the bundle fold in
// between, so the GPU folds while the host reconstructs the order (~0.18 ms each, formerly one after the other).
int merged_reference_order_begin(vbx_ctx* ctx, size_t n, uint32_t nb, const KeyFrame& kf) {
  hipStream_t s = ctx->stream;
  HIP_TRY(ctx->b_bkeys.ensure((size_t)nb * 8));   // bpack
  HIP_TRY(ctx->b_bfirst.ensure((size_t)nb * 8));  // the same in insertion order
  HIP_TRY(ctx->b_bperm.ensure((size_t)nb * 4));
  HIP_TRY(ctx->b_cnt.ensure((n + 1) * 4));        // by_s
  HIP_TRY(ctx->b_off.ensure((n + 1) * 4));        // flags
  HIP_TRY(ctx->b_T.ensure((n + 1) * 4));          // positions
  uint32_t* by_s = ctx->b_cnt.as<uint32_t>();
  HIP_TRY(hipMemsetAsync(by_s, 0xFF, n * 4, s));
  KLAUNCH(k_merged_mark_first, grid_for(n), dim3(256), 0, s, ctx->b_keys1.as<uint64_t>(),
                     ctx->b_vals1.as<uint32_t>(), ctx->b_head.as<uint32_t>(), ctx->b_rank.as<uint32_t>(),
                     (uint32_t)n, kf, by_s, ctx->b_bkeys.as<uint64_t>());
  KLAUNCH(k_merged_first_flags, grid_for(n + 1), dim3(256), 0, s, by_s, (uint32_t)n, ctx->b_off.as<uint32_t>());
  int rc = exclusive_scan_u32(ctx, ctx->b_off.as<uint32_t>(), ctx->b_T.as<uint32_t>(), n + 1);
  if (rc) return rc;
  KLAUNCH(k_merged_insertion_order, grid_for(n), dim3(256), 0, s, by_s, ctx->b_T.as<uint32_t>(),
                     ctx->b_bkeys.as<uint64_t>(), (uint32_t)n, ctx->b_bfirst.as<uint64_t>());
  HIP_TRY(ctx->h_mkeys.ensure((size_t)nb * 8));
  HIP_TRY(ctx->h_mperm.ensure((size_t)nb * 4));
  HIP_TRY(hipMemcpyAsync(ctx->h_mkeys.p, ctx->b_bfirst.p, (size_t)nb * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipEventRecord(ctx->ev_copy, s));
  return VBX_OK;
}
int merged_reference_order_finish(vbx_ctx* ctx, uint32_t nb, const uint32_t** perm_out) {
  hipStream_t s = ctx->stream;
  const uint64_t* packed = ctx->h_mkeys.as<uint64_t>();
  uint32_t* perm = ctx->h_mperm.as<uint32_t>();
  HIP_TRY(hipEventSynchronize(ctx->ev_copy));
  const auto dbg_t0 = std::chrono::steady_clock::now();
  // voxel_map (normal bundles) is walked before clear_map (tsdf_integrator.cc:324-333): split the
  // insertion sequence, each part keeps its order
  std::vector<uint32_t>&hashes = ctx->h_mhash, &order = ctx->h_morder, &idx = ctx->h_midx;
  hashes.resize(nb);
  idx.resize(nb);
  uint32_t n1 = 0;
  for (uint32_t q = 0; q < nb; ++q) n1 += !(packed[q] >> 63);
  {
    uint32_t q1 = 0, q2 = n1;
    for (uint32_t q = 0; q < nb; ++q) {
      const uint64_t v = packed[q];
      const uint32_t at = (v >> 63) ? q2++ : q1++;
      hashes[at] = (uint32_t)v;
      idx[at] = (uint32_t)(v >> 32) & 0x7FFFFFFFu;
    }
  }
  uint32_t row = 0;
  const OrderScratch sc{&ctx->h_mseq, &ctx->h_mruns, &ctx->h_by_s, &ctx->h_mnxt};
  for (int pass = 0; pass < 2; ++pass) {
    const uint32_t lo = pass ? n1 : 0, hi = pass ? nb : n1;
    unordered_iteration_order(hashes.data() + lo, hi - lo, &order, sc);
    for (uint32_t r = 0; r < hi - lo; ++r) perm[idx[lo + order[r]]] = row++;
  }
  static const bool dbg_merged = getenv("VBX_DEBUG_MERGED") != nullptr;  // read once
  if (dbg_merged)
    fprintf(stderr, "merged host order: nb=%u %.1f us\n", nb,
            std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - dbg_t0).count());
  // page-locked staging: the copy is queued and the stream goes on (the buffer is rewritten only
  // after the next frame's read-back has synchronised the stream)
  HIP_TRY(hipMemcpyAsync(ctx->b_bperm.p, perm, (size_t)nb * 4, hipMemcpyHostToDevice, s));
  *perm_out = ctx->b_bperm.as<uint32_t>();
  return VBX_OK;
}

int integrate_merged(vbx_ctx* ctx, const vbx_tsdf_cfg* cfg, const Pose& T, const float* d_pts,
                     const uint32_t* d_rgba, size_t n, int freespace) {
  CastCfg c = make_cast_cfg(ctx, cfg, &T.t.x);
  const uint32_t* order = nullptr;
  {
    int orc_ = visiting_order(ctx, cfg, d_pts, n, &order);
    if (orc_) return orc_;
  }
  hipStream_t s = ctx->stream;
  int rc = ensure_tab(ctx, false, n, false);
  if (rc) return rc;
  HIP_TRY(ctx->b_pcx.ensure(n * 4)); HIP_TRY(ctx->b_pcy.ensure(n * 4)); HIP_TRY(ctx->b_pcz.ensure(n * 4));
  RayTab pt = make_tab(ctx, false, (uint32_t)n);
  pt.bkey = nullptr;
  const uint32_t prep_blocks = grid_for(n).x;
  HIP_TRY(ctx->b_bbox.ensure((size_t)prep_blocks * 6 * 4));
  KLAUNCH(k_prep_points, grid_for(n), dim3(256), 0, s, d_pts, d_rgba, n, T, c, freespace,
                     pt, ctx->b_pcx.as<float>(), ctx->b_pcy.as<float>(), ctx->b_pcz.as<float>(), order,
                     ctx->map.voxel_size_inv, (uint64_t*)nullptr, (uint32_t*)nullptr, ctx->b_bbox.as<int32_t>(), ctx->d_state);
  KLAUNCH(k_bbox_reduce, dim3(1), dim3(256), 0, s, ctx->b_bbox.as<int32_t>(), prep_blocks, ctx->d_state);
  rc = sync_state(ctx);
  if (rc) return rc;
  // bundleRays (tsdf_integrator.cc:340-371): group points by endpoint voxel.  A stable sort
  // of (key, s) keeps each bundle's points in visiting order.  Keys are relative to the cloud's bounding
  // box: a room needs 7-8 bits per axis (two radix passes) where the absolute index takes 3 x 21 (six).
  KeyFrame kf{-(1 << 20), -(1 << 20), -(1 << 20), 21, 21, 21};  // the absolute packing (clearing bit = bit 63)
  if (!ctx->h_state.bbox_wide && ctx->h_state.bbox_min[0] <= ctx->h_state.bbox_max[0]) {
    kf.xmin = ctx->h_state.bbox_min[0]; kf.ymin = ctx->h_state.bbox_min[1]; kf.zmin = ctx->h_state.bbox_min[2];
    kf.bx = (int)bits_for((uint64_t)(ctx->h_state.bbox_max[0] - kf.xmin));
    kf.by = (int)bits_for((uint64_t)(ctx->h_state.bbox_max[1] - kf.ymin));
    kf.bz = (int)bits_for((uint64_t)(ctx->h_state.bbox_max[2] - kf.zmin));
  }
  HIP_TRY(ctx->b_keys0.ensure(n * 8)); HIP_TRY(ctx->b_keys1.ensure(n * 8));
  HIP_TRY(ctx->b_vals0.ensure(n * 4)); HIP_TRY(ctx->b_vals1.ensure(n * 4));
  KLAUNCH(k_merged_keys, grid_for(n), dim3(256), 0, s, pt, (uint32_t)n, ctx->map, kf,
                     ctx->b_keys0.as<uint64_t>(), ctx->b_vals0.as<uint32_t>());
  // one bit above the clearing bit: set only in the all-ones key of dropped points, which thereby sort last
  rc = stable_sort01(ctx, n, 0, (unsigned)std::min(64, keyframe_bits(kf) + 1), true);
  if (rc) return rc;
  HIP_TRY(ctx->b_head.ensure((n + 1) * 4)); HIP_TRY(ctx->b_rank.ensure((n + 1) * 4));
  KLAUNCH(k_merged_heads, grid_for(n + 1), dim3(256), 0, s, ctx->b_keys1.as<uint64_t>(),
                     (uint32_t)n, ctx->b_head.as<uint32_t>());
  rc = exclusive_scan_u32(ctx, ctx->b_head.as<uint32_t>(), ctx->b_rank.as<uint32_t>(), n + 1);
  if (rc) return rc;
  uint32_t nb = 0;
  rc = sync_state(ctx, ctx->b_rank.as<uint32_t>() + n, &nb);
  if (rc) return rc;
  if (nb == 0) return VBX_OK;
  rc = ensure_tab(ctx, true, nb, true);
  if (rc) return rc;
  HIP_TRY(ctx->b_graze.ensure((size_t)nb * 8));
  HIP_TRY(hipMemsetAsync(ctx->b_graze.p, 0xFF, (size_t)nb * 8, s));
  RayTab bt = make_tab(ctx, true, nb);
  const bool ref_order = cfg->merged_bundle_order == 0;
  if (ref_order) {
    rc = merged_reference_order_begin(ctx, n, nb, kf);
    if (rc) return rc;
    rc = ensure_tab_c(ctx, nb);
    if (rc) return rc;
  }
  HIP_TRY(ctx->b_bstart.ensure((size_t)nb * 4));
  HIP_TRY(ctx->b_mgather.ensure(n * 20));
  float* gw = ctx->b_mgather.as<float>();
  float* gx = gw + n;
  float* gy = gx + n;
  float* gz = gy + n;
  uint32_t* gc = reinterpret_cast<uint32_t*>(gz + n);
  KLAUNCH(k_merged_starts, grid_for(n), dim3(256), 0, s, ctx->b_head.as<uint32_t>(),
                     ctx->b_rank.as<uint32_t>(), (uint32_t)n, ctx->b_bstart.as<uint32_t>());
  KLAUNCH(k_merged_gather, grid_for(n), dim3(256), 0, s, ctx->b_keys1.as<uint64_t>(),
                     ctx->b_vals1.as<uint32_t>(), (uint32_t)n, pt, ctx->b_pcx.as<float>(), ctx->b_pcy.as<float>(),
                     ctx->b_pcz.as<float>(), gw, gx, gy, gz, gc);
  // the fold writes its rows in key order (table C) while the host works out the visiting order
  KLAUNCH(k_merged_bundle8, grid_for((size_t)nb * 8), dim3(256), 0, s, ctx->b_keys1.as<uint64_t>(),
                     ctx->b_bstart.as<uint32_t>(), nb, (uint32_t)n, gw, gx, gy, gz, gc, T, kf,
                     ref_order ? make_tab_c(ctx, nb) : bt, ctx->b_graze.as<uint64_t>(), (const uint32_t*)nullptr);
  if (ref_order) {
    const uint32_t* perm = nullptr;
    rc = merged_reference_order_finish(ctx, nb, &perm);
    if (rc) return rc;
    KLAUNCH(k_merged_permute_rows, grid_for(nb), dim3(256), 0, s, make_tab_c(ctx, nb), perm, nb, bt);
  }
  tmark(ctx, 1);
  // Non-clearing bundles sort before clearing ones (bit 63), so the graze key list is the
  // sorted prefix of non-clearing bundle keys; entries of clearing bundles stay ~0 (sorted last).
  const uint64_t* graze = c.anti_grazing ? ctx->b_graze.as<uint64_t>() : nullptr;
  return march_and_fold(ctx, bt, c, /*from_origin=*/true, nullptr, false, graze, nb, /*two_pass=*/true);
}

constexpr uint32_t kFastSetSize = (1u << 20) + 10000u;  // ApproxHashSet<20, 10000>

// Per-frame bookkeeping of FastTsdfIntegrator::integratePointCloud (tsdf_integrator.cc:564-569):
// every clear_checks_every_n_frames-th call -- empty clouds included -- both ApproxHashSets move
// on by one offset, and are zeroed when the offset wraps at 10000 (approx_hash_array.h:156-169).
int fast_frame_tick(vbx_ctx* ctx, const vbx_tsdf_cfg* cfg) {
  // (one counter per process, like the reference's function-static: ++counter >= n resets it and clears THIS integrator's sets)
  if (g_fast_reset_counter.fetch_add(1) + 1 < (int64_t)cfg->clear_checks_every_n_frames) return VBX_OK;
  hipStream_t s = ctx->stream;
  g_fast_reset_counter.store(0);
  ++ctx->obs_epoch;  // voxel_observed set cleared (exact-set form of resetApproxSet)
  if (++ctx->obsset_offset >= 10000u) {
    if (ctx->obsset_init) HIP_TRY(hipMemsetAsync(ctx->b_obsset.p, 0, (size_t)kFastSetSize * 4, s));
    ctx->obsset_offset = 0;
    ctx->obsset_sentinel_live = true;
  }
  if (++ctx->start_offset >= 10000u) {
    if (ctx->startset_init) HIP_TRY(hipMemsetAsync(ctx->b_startset.p, 0, (size_t)kFastSetSize * 4, s));
    ctx->start_offset = 0;
    ctx->start_sentinel_live = true;
  }
  return VBX_OK;
}

int integrate_fast(vbx_ctx* ctx, const vbx_tsdf_cfg* cfg, const Pose& T, const float* d_pts,
                   const uint32_t* d_rgba, size_t n, int freespace) {
  CastCfg c = make_cast_cfg(ctx, cfg, &T.t.x);
  c.take_limit = ctx->fast_take_limit;
  const uint32_t* order = nullptr;
  {
    int orc_ = visiting_order(ctx, cfg, d_pts, n, &order);
    if (orc_) return orc_;
  }
  hipStream_t s = ctx->stream;
  MapDev& m = ctx->map;
  constexpr uint32_t kSetSize = kFastSetSize;
  int rc = fast_frame_tick(ctx, cfg);
  if (rc) return rc;
  if (!ctx->startset_init) {
    HIP_TRY(ctx->b_startset.ensure((size_t)kSetSize * 4));
    HIP_TRY(hipMemsetAsync(ctx->b_startset.p, 0, (size_t)kSetSize * 4, s));
    ctx->startset_init = true;  // the offset keeps counting while the set is unallocated
  }

  rc = ensure_tab(ctx, false, n, false);
  if (rc) return rc;
  RayTab pt = make_tab(ctx, false, (uint32_t)n);
  pt.bkey = nullptr;
  HIP_TRY(ctx->b_keys0.ensure(n * 8)); HIP_TRY(ctx->b_keys1.ensure(n * 8));
  HIP_TRY(ctx->b_vals0.ensure(n * 4)); HIP_TRY(ctx->b_vals1.ensure(n * 4));
  KLAUNCH(k_prep_points, grid_for(n), dim3(256), 0, s, d_pts, d_rgba, n, T, c, freespace,
                     pt, (float*)nullptr, (float*)nullptr, (float*)nullptr, order, ctx->map.voxel_size_inv,
                     ctx->b_keys0.as<uint64_t>(), ctx->b_vals0.as<uint32_t>(), (int32_t*)nullptr, ctx->d_state);
  // keys[s] = slot << 32 | s is written in visiting order: a stable sort on the 20 slot bits
  // (+ bit 52, set only in the all-ones key of dropped points) orders by (slot, s)
  rc = stable_sort01(ctx, n, 32, 53, true);
  if (rc) return rc;
  KLAUNCH(k_fast_start_dedupe, grid_for(n), dim3(256), 0, s, ctx->b_keys1.as<uint64_t>(),
                     ctx->b_vals1.as<uint32_t>(), (uint32_t)n, ctx->b_startset.as<uint32_t>(),
                     ctx->start_offset, ctx->start_sentinel_live ? 1 : 0, pt.flags);
  HIP_TRY(ctx->b_head.ensure((n + 1) * 4)); HIP_TRY(ctx->b_rank.ensure((n + 1) * 4));
  KLAUNCH(k_fast_start_commit_and_flags, grid_for(n + 1), dim3(256), 0, s, ctx->b_keys1.as<uint64_t>(),
                     ctx->b_vals1.as<uint32_t>(), (uint32_t)n, ctx->b_startset.as<uint32_t>(),
                     ctx->start_offset, pt.flags, ctx->b_head.as<uint32_t>(), ctx->d_state);
  rc = exclusive_scan_u32(ctx, ctx->b_head.as<uint32_t>(), ctx->b_rank.as<uint32_t>(), n + 1);
  if (rc) return rc;
  rc = ensure_tab(ctx, true, n, false);
  if (rc) return rc;
  RayTab kt = make_tab(ctx, true, 0);
  kt.bkey = nullptr;
  KLAUNCH(k_compact_rays, grid_for(n), dim3(256), 0, s, pt, ctx->b_head.as<uint32_t>(),
                     ctx->b_rank.as<uint32_t>(), (uint32_t)n, kt, ctx->d_state);
  rc = sync_state(ctx);
  if (rc) return rc;
  if (ctx->h_state.sentinel_cleared) ctx->start_sentinel_live = false;
  const uint32_t R = (uint32_t)ctx->h_state.num_kept;
  kt.R = R;
  tmark(ctx, 1);
  if (R == 0) return VBX_OK;

  // Candidate blocks along the full (unterminated) paths; a block only becomes part of the
  // Layer ("published") when a ray actually reaches it (k_fast_emit).
  HIP_TRY(ctx->b_cnt.ensure((size_t)(R + 1) * 4));
  HIP_TRY(ctx->b_off.ensure((size_t)(R + 1) * 4));
  KLAUNCH(k_ray_count, grid_for(R + 1), dim3(256), 0, s, kt, c, m, 0,
                     (const uint32_t*)nullptr, ctx->b_cnt.as<uint32_t>(), 0, ctx->d_state);
  rc = exclusive_scan_u32(ctx, ctx->b_cnt.as<uint32_t>(), ctx->b_off.as<uint32_t>(), R + 1);
  if (rc) return rc;
  // Every ray emits at most sqrt(3) * (max_ray_length + truncation) / voxel_size + 4 voxels
  // (L1 <= sqrt(3) L2 of the walked segment), so the list buffer is sized without waiting for
  // the exact total; 288 GB of HBM make the slack irrelevant and the buffer is reused.
  const double seg = (double)c.max_ray_length_m + (double)c.trunc;
  const size_t per_ray = (size_t)(1.7320508075688772 * seg * (double)m.voxel_size_inv) + 6;
  if ((size_t)R * per_ray + 64 > 0xFFFFFFF0ull) {  // 32-bit list offsets
    ctx->fail("cloud too large for one call: %u rays of up to %zu voxels exceed 2^32 list entries; split the cloud", R,
              per_ray);
    return VBX_ERR_CAPACITY;
  }
  const size_t vox_cap = (size_t)R * per_ray + 64;
  HIP_TRY(ctx->b_vox.ensure(vox_cap * 4));
  HIP_TRY(ctx->b_vhash.ensure(vox_cap * 4));
  HIP_TRY(ctx->b_redo.ensure((size_t)(R + 1) * 4));
  // open list entries (blocks that get their slot after the list pass): room for 2 M of them, the frames of a moving
  // sensor produce a few thousand; beyond that the second pass below rebuilds the queued rays
  const uint32_t fix_cap = (uint32_t)std::min<size_t>(vox_cap, (size_t)2 << 20);
  HIP_TRY(ctx->b_fix.ensure((size_t)fix_cap * 12));
  auto build_lists = [&]() -> int {
    KLAUNCH(k_fast_build_lists<kListRPW>, dim3((R + 4 * kListRPW - 1) / (4 * kListRPW)), dim3(256), 0, s, kt, c, m, ctx->b_off.as<uint32_t>(),
                       ctx->b_vox.as<uint32_t>(), ctx->b_vhash.as<uint32_t>(), (uint32_t)vox_cap, ctx->b_newlist.as<uint32_t>(),
                       (const uint32_t*)nullptr, ctx->b_redo.as<uint32_t>(), ctx->d_state, ctx->b_fix.as<uint32_t>(), fix_cap);
    KLAUNCH(k_assign_slots, grid_for(m.cap_blocks), dim3(256), 0, s, m,
                       ctx->b_newlist.as<uint32_t>(), ctx->d_state);
    KLAUNCH(k_commit_alloc, dim3(1), dim3(1), 0, s, m, ctx->d_state);
    KLAUNCH(k_fast_fixup, dim3(256), dim3(256), 0, s, m, ctx->b_fix.as<uint32_t>(), fix_cap, ctx->b_vox.as<uint32_t>(), ctx->d_state);
    // second pass over the queued rays (grid sized for the first-frame worst case; idle
    // workgroups leave at once); capacity / lookup errors surface at the solver's first check
    if (ctx->fast_redo_grid == 0) ctx->fast_redo_grid = R;  // first frame: every block is new
    KLAUNCH(k_fast_build_lists<kListRPW>, dim3((std::max<uint32_t>(ctx->fast_redo_grid, 1024) + 4 * kListRPW - 1) / (4 * kListRPW)), dim3(256), 0, s, kt, c, m,
                       ctx->b_off.as<uint32_t>(), ctx->b_vox.as<uint32_t>(), ctx->b_vhash.as<uint32_t>(), (uint32_t)vox_cap,
                       ctx->b_newlist.as<uint32_t>(), ctx->b_redo.as<uint32_t>(), (uint32_t*)nullptr, ctx->d_state,
                       (uint32_t*)nullptr, 0u);
    return VBX_OK;
  };
  rc = build_lists();
  if (rc) return rc;
  tmark(ctx, 2);

  // claim arrays + tags (see k_fast_sweep)
  // ray-index bits: sized by the cloud (an upper bound of R) so the tag layout — and with it
  // the claim arrays' contents — stays valid from frame to frame
  const int s_bits = std::max((int)bits_for(std::max<size_t>(n, 2) - 1), ctx->own_s_bits);
  const bool strict_set = cfg->fast_observed_set == 0;  // the reference's ApproxHashSet semantics
  const bool keep_observed = cfg->clear_checks_every_n_frames > 1 && !strict_set;
  auto ensure_claims = [&]() -> int {  // sized by the pool: again after the pool has grown
    const size_t nvox_total = (size_t)m.cap_blocks * m.nvox;
    const bool fresh = (ctx->b_own0.p == nullptr);
    HIP_TRY(ctx->b_own0.ensure(nvox_total * 4));
    HIP_TRY(ctx->b_own1.ensure(nvox_total * 4));
    HIP_TRY(ctx->b_cl.ensure(nvox_total * 4));
    const uint32_t max_tag = (1u << (32 - s_bits)) - 2;
    if (fresh || s_bits != ctx->own_s_bits || ctx->own_tag < 1024) {
      HIP_TRY(hipMemsetAsync(ctx->b_own0.p, 0xFF, nvox_total * 4, s));
      HIP_TRY(hipMemsetAsync(ctx->b_own1.p, 0xFF, nvox_total * 4, s));
      HIP_TRY(hipMemsetAsync(ctx->b_cl.p, 0xFF, nvox_total * 4, s));
      ctx->own_s_bits = s_bits;
      ctx->own_tag = max_tag;
    }
    if (keep_observed && ctx->b_obs.cap < nvox_total * 4) {
      HIP_TRY(ctx->b_obs.ensure(nvox_total * 4));
      HIP_TRY(hipMemsetAsync(ctx->b_obs.p, 0, nvox_total * 4, s));
    }
    return VBX_OK;
  };
  rc = ensure_claims();
  if (rc) return rc;
  HIP_TRY(ctx->b_T.ensure((size_t)(R + 1) * 4));
  HIP_TRY(ctx->b_TH.ensure((size_t)(R + 1) * 4));
  HIP_TRY(ctx->b_U.ensure((size_t)(R + 1) * 4));
  HIP_TRY(ctx->b_rank.ensure((size_t)(R + 1) * 4));
  HIP_TRY(ctx->b_act0.ensure((size_t)(R + 1) * 4));
  HIP_TRY(ctx->b_act1.ensure((size_t)(R + 1) * 4));
  uint32_t iters_total = 0;
  static const uint32_t solver_sweeps = getenv("VBX_SOLVER_SWEEPS") ? (uint32_t)atoi(getenv("VBX_SOLVER_SWEEPS")) : 0u;
  static const bool solver_open_th = !(getenv("VBX_SOLVER_OPEN_GUESS") && getenv("VBX_SOLVER_OPEN_GUESS")[0] == 'l');
  auto run_solver = [&]() -> int {
    SweepArgs sa{};
    sa.off = ctx->b_off.as<uint32_t>();
    sa.vox = ctx->b_vox.as<uint32_t>();
    sa.cl = ctx->b_cl.as<uint32_t>();
    sa.tag_cl = --ctx->own_tag;
    sa.s_bits = s_bits;
    sa.max_consecutive = c.max_consecutive;
    sa.TL = ctx->b_T.as<uint32_t>();
    sa.TH = ctx->b_TH.as<uint32_t>();
    sa.U = ctx->b_U.as<uint32_t>();
    sa.obs = keep_observed ? ctx->b_obs.as<uint32_t>() : nullptr;
    sa.obs_epoch = ctx->obs_epoch;
    // k_fast_seed_claims: the first max_consecutive + 1 probes of every ray are certain (TL), TH = path length;
    // sweep 0: upper bounds from those certain claims, possible claims published up to them (h_only);
    // sweep 1: lower bounds only;
    // sweeps 2..: both bounds, open rays only.  (A persistent tail kernel with grid barriers
    // instead of launches was measured slower: 1.07 vs 0.98 ms — barrier + L2 write-back per
    // sweep cost more than a launch.)
    uint32_t iters = 0;
    uint32_t n_open = R;  // host-side upper bound of the open list
    uint32_t tag_rd = 0xFFFFFFFFu;
    int ch_flip = 0;      // which of the two possible-claim arrays holds the readable sweep
    int list_sel = 0;     // which work list the next sweep reads
    int cnt_cur = 0;      // act_count index of that list's length
    bool have_list = false;
    uint32_t* lists[2] = {ctx->b_act0.as<uint32_t>(), ctx->b_act1.as<uint32_t>()};
    uint32_t* chs[2] = {ctx->b_own0.as<uint32_t>(), ctx->b_own1.as<uint32_t>()};
    static const bool full_path_start = getenv("VBX_SOLVER_FULL_PATH_START") != nullptr;  // A/B switch: round 2's first sweep
    if (!full_path_start)
      KLAUNCH(k_fast_seed_claims, grid_for(R + 1), dim3(256), 0, s, sa.off, sa.vox, R, sa.cl, sa.tag_cl, s_bits, c.max_consecutive,
                         sa.TL, sa.TH, sa.U);
    for (;;) {
      {
        // sweeps per host check (an idle sweep is ~5 us, a check ~30 us): the first three give
        // the open-ray count that sizes the later grids; then one batch up to where the previous
        // frame converged (consecutive frames behave alike), then fours
        int kBatch = (iters == 0) ? 3 : 4;
        if (iters == 3 && ctx->fast_last_iters > 7) kBatch = (int)ctx->fast_last_iters - 3 + 1;
        if (strict_set && solver_sweeps) kBatch = (iters == 0) ? 3 : std::max(1, (int)solver_sweeps - (int)iters);
        for (int b = 0; b < kBatch; ++b) {
          sa.init = (full_path_start && iters == 0) ? 1 : 0;
          sa.h_only = (!full_path_start && iters == 0) ? 1 : 0;
          sa.sweep_idx = iters;
          sa.l_only = (iters == 1) ? 1 : 0;
          const bool writes_ch = !sa.l_only;
          const bool writes_list = iters >= 2;
          sa.cnt_in = cnt_cur;
          sa.cnt_out = (cnt_cur + 1) % 3;
          sa.list_in = have_list ? lists[list_sel] : nullptr;
          sa.list_out = writes_list ? lists[have_list ? (list_sel ^ 1) : 0] : nullptr;
          sa.n_in = n_open;
          sa.ch_rd = chs[ch_flip];
          sa.ch_wr = chs[ch_flip ^ 1];
          sa.tag_rd = tag_rd;
          sa.tag_wr = writes_ch ? --ctx->own_tag : 0;
          if (iters == 0 && full_path_start)
            KLAUNCH(k_fast_sweep<64>, grid_for((size_t)n_open * 64), dim3(256), 0, s, sa, R, ctx->d_state);
          else if (n_open <= 8192)  // few open rays: a whole wave per ray (64 list entries per step)
            KLAUNCH(k_fast_sweep<64>, grid_for((size_t)n_open * 64), dim3(256), 0, s, sa, R, ctx->d_state);
          else
            KLAUNCH(k_fast_sweep<16>, grid_for((size_t)n_open * 16), dim3(256), 0, s, sa, R, ctx->d_state);
          if (writes_ch) {
            tag_rd = sa.tag_wr;
            ch_flip ^= 1;
          }
          if (writes_list) {
            list_sel = have_list ? (list_sel ^ 1) : 0;
            have_list = true;
            cnt_cur = sa.cnt_out;
          }
          ++iters;
        }
        rc = sync_state(ctx);
        if (rc) return rc;
        rc = check_state_error(ctx);
        if (rc) return rc;
        n_open = ctx->h_state.act_count[cnt_cur];
        // size the next frame's second list-building pass (it is a grid-stride loop, so this is
        // only a performance hint)
        ctx->fast_redo_grid = std::max<uint32_t>(1, 2 * ctx->h_state.redo_count);
      }
      static const bool dbg_solver = getenv("VBX_DEBUG") != nullptr;
      if (dbg_solver) fprintf(stderr, "[vbx] fast solver: after %u sweeps %u open rays of %u (redo rays %u, blocks published so far %u, pool used %u)\n", iters, n_open, R,
                              ctx->h_state.redo_count, ctx->h_state.blocks_published, ctx->h_state.pool_used);
      if (n_open == 0) break;
      // The reference's approximate set: the exact-set solution is only the replay's first guess, and the replay
      // converges to the sequential result from ANY guess.  Past a few sweeps the solver is refining a handful of rays
      // at 23 us per launch; the replay settles them in the rounds it runs anyway.  Open rays enter with their upper
      // bound (surplus probes vanish in one round; a guess that ends early grows over several).
      if (strict_set && solver_sweeps && iters >= solver_sweeps) {
        if (solver_open_th && have_list)
          KLAUNCH(k_fast_open_guess, grid_for(n_open), dim3(256), 0, s, lists[list_sel], &ctx->d_state->act_count[cnt_cur], n_open,
                  sa.TL, sa.TH);
        break;
      }
      if (iters > 1000000 || ctx->own_tag < 128) {
        ctx->fail("Fast integrator: early-termination solver did not converge");
        return VBX_ERR_HIP;
      }
    }
    // sweeps that did work (the launches after convergence are idle)
    if (ctx->h_state.fast_idle_sweep) iters = std::min(iters, 0xFFFFFFFFu - ctx->h_state.fast_idle_sweep);
    iters_total = iters;
    ctx->fast_last_iters = std::min<uint32_t>(iters, 64);
    return VBX_OK;
  };
  for (;;) {
    rc = run_solver();
    if (rc != VBX_ERR_CAPACITY || !(ctx->h_state.error & 1u)) break;
    // the pool ran out of slots while the lists were built (Layer::allocateBlockPtrByIndex never fails,
    // layer.h:133-160): double it, build the lists again, restart the solver
    rc = grow_pool(ctx);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(&ctx->d_state->act_count[0], 0, 12, s));
    HIP_TRY(hipMemsetAsync(&ctx->d_state->redo_count, 0, 12, s));  // + fix_count, fix_overflow
    HIP_TRY(hipMemsetAsync(&ctx->d_state->fast_idle_sweep, 0, 4, s));
    rc = build_lists();
    if (rc) return rc;
    rc = ensure_claims();
    if (rc) return rc;
  }
  if (rc) return rc;
  if (keep_observed)
    KLAUNCH(k_fast_mark_observed, grid_for((size_t)R * 16), dim3(256), 0, s, ctx->b_off.as<uint32_t>(),
                       ctx->b_vox.as<uint32_t>(), ctx->b_T.as<uint32_t>(), R, ctx->b_obs.as<uint32_t>(),
                       ctx->obs_epoch);
  if (strict_set) {
    tmark(ctx, 8);
    // refinement rounds (see k_strict_keys): T lives in b_T (updated in place), probe offsets in b_cnt
    if (!ctx->obsset_init) {
      HIP_TRY(ctx->b_obsset.ensure((size_t)kSetSize * 4));
      HIP_TRY(hipMemsetAsync(ctx->b_obsset.p, 0, (size_t)kSetSize * 4, s));
      ctx->obsset_init = true;
    }
    uint32_t* Tcur = ctx->b_T.as<uint32_t>();
    uint32_t* poff = ctx->b_cnt.as<uint32_t>();
    HIP_TRY(ctx->b_moved.ensure((size_t)R + 1));
    HIP_TRY(hipMemsetAsync(&ctx->d_state->sentinel_cleared, 0, 4, s));
    uint32_t rounds = 0;
    static const bool dbg = getenv("VBX_DEBUG") != nullptr;
    // Replay of the rays [a, b) against the set content left by everything before them (the
    // persistent array: what the frame found, plus the commits of the rays below a).  Runs
    // rounds until no probe count in the range moves (then commits the range's probes) or about
    // max_rounds are used up.  Rounds are queued in BATCHES without a host check in between: the
    // number of probes of a round is only known on the device (k_strict_keys publishes it, the sort
    // and the outcome kernel read it there), buffers and grids are sized for a bound taken from the
    // last check (probe counts move by a few percent per round); a round that outgrows the bound
    // raises rp_overflow, the rest of the batch idles and the host enlarges the bound.  A round
    // after convergence reproduces the same state, so running past it is harmless.  Per round this
    // saves the read-back and the pipeline bubble behind it (~15-20 us of ~100).
    HIP_TRY(ctx->b_collided.ensure(vox_cap));  // outcomes by global probe index (< list entries)
    static const uint32_t grow_mult = getenv("VBX_GROW") ? (uint32_t)atoi(getenv("VBX_GROW")) : 4u;  // experiment switch
    auto replay = [&](uint32_t a, uint32_t b, uint32_t max_rounds, uint32_t p_hint, bool* converged) -> int {
      *converged = false;
      uint32_t done = 0;  // rounds queued so far in this call
      uint32_t bound = 0;
      // probe offsets of the rays in play: a block of rays only needs its own (offsets from the block's first probe;
      // every kernel of a round takes them relative to poff[a])
      const bool whole = (a == 0 && b == R);
      auto one_scan = [&]() -> int {
        return whole ? exclusive_scan_u32(ctx, Tcur, poff, R + 1) : exclusive_scan_u32(ctx, Tcur + a, poff + a, b - a + 1);
      };
      if (p_hint == 0) {  // no estimate of the range's probe count: ask
        rc = one_scan();
        if (rc) return rc;
        const uint32_t* const ptrs[3] = {poff + b, poff + a, poff + b};
        uint32_t vals[3] = {0, 0, 0};
        rc = sync_state3(ctx, ptrs, vals);
        if (rc) return rc;
        if (whole) ctx->h_poff_total = vals[0];
        p_hint = vals[2] - vals[1];
      }
      bound = p_hint + p_hint / 4 + 8192;
      HIP_TRY(hipMemsetAsync(&ctx->d_state->rp_overflow, 0, 8, s));  // rp_overflow, rp_changed_round
      static const uint32_t first_batch = getenv("VBX_REPLAY_FIRST_BATCH") ? (uint32_t)std::max(1, atoi(getenv("VBX_REPLAY_FIRST_BATCH"))) : 1u;  // measurement switch
      uint32_t batch = (a == 0 && b == R) ? first_batch : 1u;
      for (;;) {
        const uint32_t first = done;
        for (uint32_t q = 0; q < batch; ++q, ++done) {
          rc = one_scan();
          if (rc) return rc;
          uint32_t n_sort = bound;
          HIP_TRY(ctx->b_keys0.ensure((size_t)std::max<uint32_t>(bound, 1) * 8));
          HIP_TRY(ctx->b_keys1.ensure((size_t)std::max<uint32_t>(bound, 1) * 8));
          KLAUNCH(k_strict_keys, grid_for((size_t)(b - a) * 16), dim3(256), 0, s, poff, a, b, bound, ctx->b_off.as<uint32_t>(),
                             ctx->b_vhash.as<uint32_t>(), ctx->b_keys0.as<uint64_t>(), ctx->d_state);
          rc = stable_sort01(ctx, std::max<uint32_t>(n_sort, 1), 44, 64, false, &ctx->d_state->rp_n);
          if (rc) return rc;
          KLAUNCH(k_strict_outcome, grid_for(std::max<uint32_t>(n_sort, 1)), dim3(256), 0, s, ctx->b_keys1.as<uint64_t>(), ctx->d_state,
                             ctx->b_obsset.as<uint32_t>(), ctx->obsset_offset, ctx->obsset_sentinel_live ? 1 : 0,
                             ctx->b_collided.as<uint8_t>());
          KLAUNCH(k_strict_scan, grid_for((size_t)(b - a) * 16), dim3(256), 0, s, poff, ctx->b_off.as<uint32_t>(),
                             R, a, b, ctx->b_collided.as<uint8_t>(), c.max_consecutive, Tcur,
                             ctx->b_U.as<uint32_t>(), ctx->b_moved.as<uint8_t>(), done, grow_mult, ctx->d_state);
          ++rounds;
        }
        rc = sync_state(ctx);
        if (rc) return rc;
        if (ctx->h_state.sentinel_cleared) ctx->obsset_sentinel_live = false;
        if (a == 0 && b == R) ctx->h_poff_total = ctx->h_state.rp_n;
        if (dbg)
          fprintf(stderr, "[vbx] replay rays [%u,%u): %u rounds queued, last change in round %u, %u probes (bound %u)%s\n", a, b,
                  done, ctx->h_state.rp_changed_round, ctx->h_state.rp_n, bound, ctx->h_state.rp_overflow ? " OVERFLOW" : "");
        if (ctx->h_state.rp_overflow) {  // the batch stopped short: more room, same state
          bound = std::max<uint32_t>(2 * bound, ctx->h_state.rp_n + ctx->h_state.rp_n / 4 + 8192);
          HIP_TRY(hipMemsetAsync(&ctx->d_state->rp_overflow, 0, 4, s));
          continue;
        }
        if (ctx->h_state.rp_changed_round < done) {  // the last round(s) moved nothing
          *converged = true;
          break;
        }
        if (done >= max_rounds) break;
        bound = std::max(bound, ctx->h_state.rp_n + ctx->h_state.rp_n / 4 + 8192);
        (void)first;
        // measured: a check costs about as much as one idle round (the rounds are bound by their ~10
        // dependent kernels, not by the read-back), so batches stay short: 1, 1, 1, then pairs
        // a block of the fine-voxel replay needs a few dozen rounds (each fixes the next link of its dependency chains):
        // there the checks are spread further (measured at 0.02 m: rounds 187 -> see HISTORY.md §4.3)
        static const uint32_t blk_batch = getenv("VBX_REPLAY_BATCH") ? (uint32_t)atoi(getenv("VBX_REPLAY_BATCH")) : 2u;
        if (a != 0 || b != R) batch = done >= 2 ? blk_batch : 1;
        else batch = done >= 3 ? 2 : 1;
      }
      ctx->rp_last_p = ctx->h_state.rp_n;
      // converged: the sorted probe list of the last round (whose T equals the final T) is still
      // in keys1 — its last probe per slot is the set's content after ray b - 1
      if (*converged && ctx->h_state.rp_n)
        KLAUNCH(k_strict_commit, grid_for(ctx->h_state.rp_n), dim3(256), 0, s, ctx->b_keys1.as<uint64_t>(),
                           ctx->b_obsset.as<uint32_t>(), ctx->obsset_offset, ctx->d_state);
      return VBX_OK;
    };
    // Phase 1: the whole frame at once.  Steady-state frames converge in 1-5 rounds; a round
    // costs ~45 us + 0.07 us per 1000 probes.
    bool converged = false;
    // whole-frame rounds; the previous frame's probe count sizes the first batch (no read-back needed)
    rc = replay(0, R, 7, ctx->rp_last_p, &converged);
    if (rc) return rc;
    // Blocks pay off only when whole-frame rounds are expensive (millions of probes, i.e. fine
    // voxels): 16-32 blocks cost at least two cheap rounds each.
    static const bool no_blocks = getenv("VBX_REPLAY_NO_BLOCKS") != nullptr;  // measurement switch
    static const uint32_t blocks_min = getenv("VBX_REPLAY_BLOCKS_MIN") ? (uint32_t)atoi(getenv("VBX_REPLAY_BLOCKS_MIN")) : 1500000u;  // measurement switch
    const bool use_blocks = ctx->h_poff_total > blocks_min && !no_blocks;
    if (!converged && !use_blocks) {
      rc = replay(0, R, 100000, ctx->rp_last_p, &converged);
      if (rc) return rc;
      if (!converged) {
        ctx->fail("Fast integrator: observed-set replay did not converge");
        return VBX_ERR_HIP;
      }
    }
    const uint32_t rounds_whole_frame = rounds;
    if (!converged) {
      // Phase 2: long dependency chains (a fresh map, or more probed voxels than set slots: every
      // round then only fixes the next link).  Rays below the lowest one that still moved are
      // final: their probes are committed to the set, and the rest is replayed in blocks of
      // consecutive rays — a block only interacts with itself and with the set content left by
      // the rays before it, so its rounds are cheap (a 1/16 of the probes) and short chains
      // inside a block converge in a few of them.
      HIP_TRY(hipMemsetAsync(&ctx->d_state->act_count[2], 0xFF, 4, s));
      KLAUNCH(k_strict_first_moved, grid_for(R), dim3(256), 0, s, ctx->b_moved.as<uint8_t>(), R,
                         &ctx->d_state->act_count[2]);
      rc = exclusive_scan_u32(ctx, Tcur, poff, R + 1);
      if (rc) return rc;
      rc = sync_state(ctx);
      if (rc) return rc;
      const uint32_t r_lo = std::min(ctx->h_state.act_count[2], R);
      std::vector<uint32_t>& hp = ctx->h_poff;
      hp.resize((size_t)R + 1);
      HIP_TRY(hipMemcpyAsync(hp.data(), poff, ((size_t)R + 1) * 4, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      if (r_lo > 0) {
        rc = replay(0, r_lo, 8, hp[r_lo], &converged);  // final already: one round to list the probes, one to confirm
        if (rc) return rc;
        if (!converged) {
          ctx->fail("Fast integrator: observed-set replay: settled prefix moved again");
          return VBX_ERR_HIP;
        }
      }
      static const uint32_t kBlocks = getenv("VBX_REPLAY_BLOCKS") ? (uint32_t)atoi(getenv("VBX_REPLAY_BLOCKS")) : 16u;  // measurement switch
      const uint64_t p_lo = hp[r_lo], p_hi = hp[R];
      uint32_t a = r_lo;
      for (uint32_t j = 1; j <= kBlocks && a < R; ++j) {
        uint32_t b = R;
        if (j < kBlocks) {  // block ends where the j-th share of the remaining probes ends
          const uint64_t target = p_lo + (p_hi - p_lo) * j / kBlocks;
          b = (uint32_t)(std::lower_bound(hp.begin() + a + 1, hp.begin() + R, (uint32_t)target) - hp.begin());
          b = std::max(b, a + 1);
        }
        rc = replay(a, b, 100000, hp[b] - hp[a], &converged);
        if (rc) return rc;
        if (!converged) {
          ctx->fail("Fast integrator: observed-set replay did not converge");
          return VBX_ERR_HIP;
        }
        a = b;
      }
    }
    ctx->counters.replay_rounds = rounds;
    ctx->counters.replay_block_rounds = use_blocks ? rounds - rounds_whole_frame : 0;
  }
  uint32_t total = 0;
  // offsets of the keys each ray emits
  rc = exclusive_scan_u32(ctx, ctx->b_U.as<uint32_t>(), ctx->b_rank.as<uint32_t>(), R + 1);
  if (rc) return rc;
  rc = sync_state(ctx, ctx->b_rank.as<uint32_t>() + R, &total);
  if (rc) return rc;
  if (strict_set && ctx->h_state.sentinel_cleared) ctx->obsset_sentinel_live = false;
  ctx->counters.iterations = iters_total;
  tmark(ctx, 3);
  if (total == 0) return VBX_OK;
  HIP_TRY(ctx->b_keys0.ensure((size_t)total * 8));
  HIP_TRY(ctx->b_keys1.ensure((size_t)total * 8));
  KLAUNCH(k_fast_emit, grid_for(total), dim3(256), 0, s, ctx->b_off.as<uint32_t>(),
                     ctx->b_vox.as<uint32_t>(), ctx->b_rank.as<uint32_t>(), R, total, m,
                     ctx->b_keys0.as<uint64_t>(), ctx->d_state);
  tmark(ctx, 4);
  return sort_and_fold(ctx, kt, c, total, /*giant_runs=*/false);
}

// ---- the order in which the reference's Layer receives its blocks -------------------------------------------------------
// k_collect_new appends the blocks a call published to a device log; nothing is read back until somebody needs the order.
int drain_new_blocks(vbx_ctx* ctx);
int collect_new_blocks(vbx_ctx* ctx) {
  hipStream_t s = ctx->stream;
  const uint32_t used = std::min(ctx->h_state.pool_used, ctx->map.cap_blocks);
  const size_t cap = (size_t)ctx->map.cap_blocks + 1024;
  if (ctx->b_newlog.cap < cap * sizeof(NewLogEntry)) {
    int rc = drain_new_blocks(ctx);   // (the log moves: take what it holds first)
    if (rc) return rc;
    HIP_TRY(ctx->b_newlog.ensure(cap * sizeof(NewLogEntry)));
  }
  const uint32_t log_cap = (uint32_t)(ctx->b_newlog.cap / sizeof(NewLogEntry));
  if ((size_t)ctx->newlog_pending + ctx->h_state.blocks_published > log_cap) {
    int rc = drain_new_blocks(ctx);
    if (rc) return rc;
  }
  if (used) KLAUNCH(k_collect_new, grid_for(used), dim3(256), 0, s, ctx->map, used, ctx->b_newlog.as<NewLogEntry>(), log_cap, ctx->call_seq, ctx->d_state);
  ctx->newlog_pending += ctx->h_state.blocks_published;
  ctx->new_flags_live = false;
  ++ctx->call_seq;
  return VBX_OK;
}

// Log -> temp_block_map_ -> Layer: per call (and per pass of the Merged integrator) the new blocks are emplaced in first-touch
// order into a container keyed like the reference's (same hash, same libstdc++: same iteration order, including the bucket
// array clear() leaves behind, tsdf_integrator.cc:146), and that container's iteration order is the sequence of
// Layer::insertBlock calls (:141-144).
int drain_new_blocks(vbx_ctx* ctx) {
  if (ctx->newlog_pending == 0) return VBX_OK;
  hipStream_t s = ctx->stream;
  HIP_TRY(hipStreamSynchronize(s));
  uint32_t cnt[2] = {0, 0};
  HIP_TRY(hipMemcpy(cnt, &ctx->d_state->newlog_count, 8, hipMemcpyDeviceToHost));
  const uint32_t log_cap = (uint32_t)(ctx->b_newlog.cap / sizeof(NewLogEntry));
  const uint32_t n = std::min(cnt[0], log_cap);
  std::vector<NewLogEntry> log(n);
  if (n) HIP_TRY(hipMemcpy(log.data(), ctx->b_newlog.p, (size_t)n * sizeof(NewLogEntry), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemsetAsync(&ctx->d_state->newlog_count, 0, 8, s));
  ctx->newlog_pending = 0;
  if (cnt[1]) ctx->layer_order_exact = false;   // (entries were dropped: those blocks are listed behind the known ones)
  std::sort(log.begin(), log.end(), [](const NewLogEntry& a, const NewLogEntry& b) {
    return a.seq != b.seq ? a.seq < b.seq : a.rank < b.rank;
  });
  size_t i = 0;
  while (i < log.size()) {
    // one batch = one updateLayerWithStoredBlocks: a call, or one of the Merged integrator's two passes
    const unsigned long long seq = log[i].seq, pass = log[i].rank >> 62;
    const bool last_call = seq == ctx->last_call_seq;
    if (last_call && ctx->last_new_seq != seq) {   // the first batch of the last call: last_new starts over
      ctx->last_new.clear();
      ctx->last_new_seq = seq;
    }
    size_t j = i;
    for (; j < log.size() && log[j].seq == seq && (log[j].rank >> 62) == pass; ++j) {
      int x, y, z;
      unpack_block_key(log[j].key, &x, &y, &z);
      ctx->temp_block_map.emplace(HostBlockIdx{x, y, z}, 0);
      ctx->layer_order.erase(HostBlockIdx{x, y, z});   // (published anew: whatever the key meant before was removed since)
    }
    for (const auto& kv : ctx->temp_block_map) {
      ctx->layer_order.emplace(kv.first, 0u);
      if (last_call) ctx->last_new.push_back(kv.first);
    }
    ctx->temp_block_map.clear();
    i = j;
  }
  return VBX_OK;
}

// Every block of the layer goes (vbx_clear, vbx_clear_keep_slots): whatever the log still holds names blocks that go too —
// dropped on the device, nothing is read back (a delta map of the sharding is cleared every step).
int discard_new_blocks(vbx_ctx* ctx) {
  // A map that follows the Layer's order replays the pending insertions first: temp_block_map_ and block_map_ keep the
  // bucket arrays those insertions grew (clear() keeps the buckets, layer.h:164, tsdf_integrator.cc:146), and where later
  // insertions land depends on them.  A map with tracking off (the sharding's delta maps) never has a pending log.
  if (ctx->newlog_pending && ctx->track_block_order) {
    int rc = drain_new_blocks(ctx);
    if (rc) return rc;
  }
  if (ctx->newlog_pending) HIP_TRY(hipMemsetAsync(&ctx->d_state->newlog_count, 0, 8, ctx->stream));
  ctx->newlog_pending = 0;
  ctx->layer_order.clear();   // block_map_.clear() (layer.h:168): the keys go, the bucket array stays
  ctx->last_new.clear();
  return VBX_OK;
}

// A list of (key, slot) pairs of published TSDF blocks -> the sequence in which the reference's Layer would iterate
// over them (Layer::getAllAllocatedBlocks / getAllUpdatedBlocks walk block_map_, layer.h:184-203).  Blocks whose insertion
// the library did not see in the reference's sequence (merged in from another map, a log that overflowed) follow in
// ascending key order and *exact turns false.
int order_like_layer(vbx_ctx* ctx, std::vector<std::pair<uint64_t, uint32_t>>* v, bool* exact, size_t* n_unknown = nullptr) {
  int rc = drain_new_blocks(ctx);
  if (rc) return rc;
  std::unordered_map<uint64_t, uint32_t> slot_of;
  slot_of.reserve(v->size() * 2);
  for (const auto& kv : *v) slot_of.emplace(kv.first, kv.second);
  std::vector<std::pair<uint64_t, uint32_t>> out;
  out.reserve(v->size());
  for (const auto& kv : ctx->layer_order) {
    const uint64_t key = pack_block_key(kv.first.x, kv.first.y, kv.first.z);
    const auto it = slot_of.find(key);
    if (it == slot_of.end()) continue;
    out.emplace_back(key, it->second);
    slot_of.erase(it);
  }
  bool ex = ctx->layer_order_exact;
  if (n_unknown) *n_unknown = slot_of.size();
  if (!slot_of.empty()) {
    ex = false;
    std::vector<std::pair<uint64_t, uint32_t>> rest(slot_of.begin(), slot_of.end());
    std::sort(rest.begin(), rest.end());
    out.insert(out.end(), rest.begin(), rest.end());
  }
  v->swap(out);
  if (exact) *exact = ex;
  return VBX_OK;
}

int integrate_device(vbx_ctx* ctx, int kind, const vbx_tsdf_cfg* cfg, const float pos[3],
                     const float quat[4], const float* d_pts, const uint8_t* d_rgba, size_t n,
                     int freespace) {
  if (!cfg || !pos || !quat || (n && (!d_pts || !d_rgba))) {
    ctx->fail("vbx_tsdf_integrate: null argument");
    return VBX_ERR_INVALID;
  }
  if (n >= (1ull << 31)) {
    ctx->fail("vbx_tsdf_integrate: too many points");
    return VBX_ERR_INVALID;
  }
  if (cfg->integration_order_mode != 0 && cfg->integration_order_mode != 1) {
    ctx->fail("Unknown integration order mode");  // integrator_utils.cc:12
    return VBX_ERR_INVALID;
  }
  if (kind != VBX_TSDF_SIMPLE && kind != VBX_TSDF_MERGED && kind != VBX_TSDF_FAST) {
    ctx->fail("unknown TSDF integrator type %d", kind);  // tsdf_integrator.cc:40-43
    return VBX_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  ctx->counters = vbx_counters{};
  ctx->counters.points = n;
  if (n == 0) return kind == VBX_TSDF_FAST ? fast_frame_tick(ctx, cfg) : VBX_OK;
  // FastTsdfIntegrator stops taking points once max_integration_time_s of wall clock are used up
  // (tsdf_integrator.cc:496-499; Simple and Merged have no such check).  A budget that is already spent at the
  // first point (<= 0, NaN) integrates nothing, exactly like the reference.  A positive one: the device path cannot
  // look at the clock half way through a frame, so it decides BEFORE the frame how many points the budget pays for,
  // from the time per taken point of this handle's earlier calls, and takes that prefix of the reference's taking
  // order (ThreadSafeIndex: mixed or sorted) — the points a single-threaded reference would have got to.  The first
  // call of a handle has no measurement and integrates everything; counters.points_taken / time_budget_exceeded report
  // what happened.
  const bool has_budget = kind == VBX_TSDF_FAST && cfg->max_integration_time_s < 1.0e30f;
  if (kind == VBX_TSDF_FAST && !(cfg->max_integration_time_s * 1000000.0f > 0.0f)) return fast_frame_tick(ctx, cfg);
  ctx->fast_take_limit = ~0u;
  ctx->counters.points_taken = n;
  if (has_budget && ctx->fast_us_per_point > 0.0) {
    double can = (double)cfg->max_integration_time_s * 1000000.0 / ctx->fast_us_per_point;
    // (the estimate is a wall-time average: the number of points taken moves by at most a factor of two from one frame
    // to the next, so that one slow frame — a host hiccup — does not cut the next one to a sliver)
    if (ctx->fast_prev_taken > 0) can = std::min(std::max(can, 0.5 * (double)ctx->fast_prev_taken), 2.0 * (double)ctx->fast_prev_taken);
    if (can < (double)n) {
      ctx->fast_take_limit = (uint32_t)can;
      ctx->counters.points_taken = ctx->fast_take_limit;
      ctx->counters.time_budget_exceeded = 1;
      if (ctx->fast_take_limit == 0) {   // (the estimate decays so that a later frame gets another try)
        ctx->fast_us_per_point *= 0.75;
        return fast_frame_tick(ctx, cfg);
      }
    }
  }
  const auto t_call0 = std::chrono::steady_clock::now();
  const uint32_t pool_grown_before = ctx->pool_grown;
  if (ctx->new_flags_live) {   // an earlier call failed half way: its new-block marks must not count for this one
    // (the host copy of the state is the failed call's last read-back: slots it allocated later would be missed, so the
    // flush reads the state first and walks the whole pool; what the failed call published enters the order behind the
    // known blocks, and the library says so: exact = 0)
    int rcn = sync_state(ctx);
    if (rcn) return rcn;
    ctx->h_state.pool_used = ctx->map.cap_blocks;
    ctx->h_state.blocks_published = std::max(ctx->h_state.blocks_published, 1u);
    rcn = collect_new_blocks(ctx);
    if (rcn) return rcn;
    rcn = sync_state(ctx);
    if (rcn) return rcn;
    ctx->layer_order_exact = false;
  }
  ctx->new_flags_live = ctx->track_block_order;
  // per-call device counters
  KLAUNCH(k_reset_call_state, dim3(1), dim3(1), 0, ctx->stream, ctx->d_state);
  Pose T;
  T.t = {pos[0], pos[1], pos[2]};
  T.qw = quat[0]; T.qx = quat[1]; T.qy = quat[2]; T.qz = quat[3];
  for (int i = 0; i < 9; ++i) ctx->ev_hit[i] = false;
  tmark(ctx, 0);
  int rc;
  const uint32_t* rgba32 = reinterpret_cast<const uint32_t*>(d_rgba);
  switch (kind) {
    case VBX_TSDF_SIMPLE: rc = integrate_simple(ctx, cfg, T, d_pts, rgba32, n, freespace); break;
    case VBX_TSDF_MERGED: rc = integrate_merged(ctx, cfg, T, d_pts, rgba32, n, freespace); break;
    default: rc = integrate_fast(ctx, cfg, T, d_pts, rgba32, n, freespace); break;
  }
  if (rc) return rc;
  tmark(ctx, 7);
  rc = sync_state(ctx);
  if (rc) return rc;
  rc = check_state_error(ctx);
  if (rc) return rc;
  ctx->counters.rays_cast = kind == VBX_TSDF_FAST ? ctx->h_state.num_kept : ctx->h_state.rays_cast;  // Fast casts every kept ray
  ctx->counters.voxels_touched = 0;
  for (int i = 0; i < 64; ++i) ctx->counters.voxels_touched += ctx->h_state.voxels_touched[i];
  ctx->counters.blocks_allocated = ctx->h_state.blocks_published;
  // the blocks this call added to the Layer, with their first-touch ranks -> the new-block log (no read-back here)
  ctx->last_call_seq = ctx->call_seq;
  ctx->published_since_clear += ctx->h_state.blocks_published;
  if (ctx->h_state.blocks_published > 0 && ctx->track_block_order) {
    rc = collect_new_blocks(ctx);
    if (rc) return rc;
  } else {
    ctx->new_flags_live = false;
    ++ctx->call_seq;
  }
  if (has_budget) {
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_call0).count();
    const double per = us / (double)std::max<uint64_t>(ctx->counters.points_taken, 1);
    // the handle's first integrate call and any call that grew the pool pay one-off costs (allocations, code objects,
    // the pool copy): they are no measure of what a point costs, and the calls after them take everything again
    const bool one_off = ctx->fast_budget_calls++ == 0 || ctx->pool_grown != pool_grown_before;
    ctx->fast_prev_taken = (uint32_t)std::min<uint64_t>(ctx->counters.points_taken, 0xFFFFFFFFu);
    if (!one_off) ctx->fast_us_per_point = ctx->fast_us_per_point > 0.0 ? 0.5 * ctx->fast_us_per_point + 0.5 * per : per;
    if (!(us < (double)cfg->max_integration_time_s * 1000000.0)) {
      ctx->counters.time_budget_exceeded = 1;
      if (!ctx->warned_time_budget) {
        ctx->warned_time_budget = true;
        fprintf(stderr, "[vbx] FastTsdfIntegrator: the frame took %.0f us, max_integration_time_s allows %.0f us; later "
                        "frames take as many points as the budget pays for at the measured rate "
                        "(reported once; see vbx_counters.points_taken / time_budget_exceeded)\n", us,
                (double)cfg->max_integration_time_s * 1000000.0);
      }
    }
  }
#ifdef VBX_FOLD_STATS  // measurement build (tools/fold_stats.py): chunks of k_fold_long by case
  ctx->counters.iterations = ctx->h_state.act_count[0];
  ctx->counters.esdf_blocks = ctx->h_state.act_count[1];
  ctx->counters.esdf_relaxations = ctx->h_state.act_count[2];
  ctx->counters.esdf_sweeps = ctx->h_state.fold_long_count[0];
  fprintf(stderr, "fold dbg:");
  for (int i = 0; i < 16; ++i) fprintf(stderr, " %u", ctx->h_state.dbg[i]);
  fprintf(stderr, "\n");
#endif
  if (ctx->timing) {
    (void)hipEventSynchronize(ctx->ev[7]);  // the state read-back spins on mapped memory; the runtime may not have retired the events yet
    float t[8] = {0};
    int last = 0;
    for (int i = 1; i < 8; ++i) {  // a stage a path skips reads as zero-length
      if (!ctx->ev_hit[i]) continue;
      (void)hipEventElapsedTime(&t[i], ctx->ev[last], ctx->ev[i]);
      last = i;
    }
    vbx_timing& o = ctx->last_timing;
    o.prep_ms = t[1]; o.alloc_ms = t[2]; o.solve_ms = t[3]; o.emit_ms = t[4];
    o.sort_ms = t[5]; o.fold_ms = t[6];
    o.replay_ms = 0.0f;
    if (ctx->ev_hit[8] && ctx->ev_hit[3]) {  // stage 3 = exact-set solve + reference-set replay rounds
      (void)hipEventElapsedTime(&o.replay_ms, ctx->ev[8], ctx->ev[3]);
      o.solve_ms = t[3] - o.replay_ms;
    }
    (void)hipEventElapsedTime(&o.total_ms, ctx->ev[0], ctx->ev[7]);
  }
  return VBX_OK;
}


}  // namespace

