// vbx_host_common.hpp — host helpers: state read-back, ray tables, sort / scan drivers (vbx_sort.hpp), stage timing
// Part of libvbx_hip.so's single translation unit (included by vbx_hip.hip, in order).

namespace {

inline dim3 grid_for(size_t n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }

// Reads DevState (and optionally one more device word) back to the host; on return everything
// queued on the stream before the call has completed.
int sync_state3(vbx_ctx* ctx, const uint32_t* const d_extra[3], uint32_t extra_out[3]) {
  if (!ctx->h_mirror) {
    HIP_TRY(hipMemcpyAsync(&ctx->h_state, ctx->d_state, sizeof(DevState), hipMemcpyDeviceToHost, ctx->stream));
    for (int i = 0; i < 3; ++i)
      if (d_extra[i]) HIP_TRY(hipMemcpyAsync(&extra_out[i], d_extra[i], 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VBX_OK;
  }
  const uint32_t seq = ++ctx->sync_seq;
  KLAUNCH(k_publish_state, dim3(1), dim3(64), 0, ctx->stream, ctx->d_state, ctx->d_mirror, d_extra[0],
                     d_extra[1], d_extra[2], seq);
  HIP_TRY(hipGetLastError());
  const auto t0 = std::chrono::steady_clock::now();
  uint32_t spins = 0;
  while (__atomic_load_n(&ctx->h_mirror->seq, __ATOMIC_ACQUIRE) != seq) {
    if ((++spins & 0xFFFu) == 0 &&
        std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
      // long-running or failed work: block, and let the runtime report an error if there is one
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      if (__atomic_load_n(&ctx->h_mirror->seq, __ATOMIC_ACQUIRE) != seq) {
        ctx->fail("device state read-back did not arrive");
        return VBX_ERR_HIP;
      }
      break;
    }
  }
  std::memcpy(&ctx->h_state, &ctx->h_mirror->st, sizeof(DevState));
  for (int i = 0; i < 3; ++i)
    if (d_extra[i]) extra_out[i] = ctx->h_mirror->extra[i];
  return VBX_OK;
}
int sync_state(vbx_ctx* ctx, const uint32_t* d_extra = nullptr, uint32_t* extra_out = nullptr) {
  const uint32_t* const ptrs[3] = {d_extra, nullptr, nullptr};
  uint32_t out[3] = {0, 0, 0};
  const int rc = sync_state3(ctx, ptrs, out);
  if (!rc && d_extra) *extra_out = out[0];
  return rc;
}

int check_state_error(vbx_ctx* ctx) {
  if (ctx->h_state.error & 1u) {
    ctx->fail("block pool / hash map capacity exceeded (max_blocks=%u)", ctx->map.cap_blocks);
    return VBX_ERR_CAPACITY;
  }
  if (ctx->h_state.error & 2u) {
    ctx->fail("internal: ray march hit a block without a pool slot");
    return VBX_ERR_HIP;
  }
  if (ctx->h_state.error & 8u) {
    ctx->fail("point cloud reaches beyond +-2^20 voxels of the map origin (%.0f m at this voxel size): "
              "voxel keys would alias", 1048576.0 * ctx->map.voxel_size);
    return VBX_ERR_INVALID;
  }
  if (ctx->h_state.error & 4u) {
    ctx->fail("internal: voxel list capacity bound violated");
    return VBX_ERR_HIP;
  }
  if (ctx->h_state.error & 64u) {
    ctx->fail("internal: fused radix pass gave up waiting for an earlier tile");
    return VBX_ERR_HIP;
  }
  return VBX_OK;
}

// Layer::allocateBlockPtrByIndex never fails (layer.h:133-160); the pool here starts at max_blocks and
// doubles when a call runs out of slots: new arrays, the used slots copied, the hash table rebuilt from
// the pool.  Keys inserted by the failed allocation pass are dropped with the old table — the caller
// re-runs that pass.  Bounded by vbx_set_pool_limit, by 2^32 voxel ids and by device memory.
static hipError_t grow_buf(DBuf& b, size_t new_bytes, size_t keep_bytes, int fill, hipStream_t s) {
  DBuf n;
  hipError_t e = n.ensure(new_bytes);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(n.p, fill, n.cap, s);
  if (e == hipSuccess && keep_bytes) e = hipMemcpyAsync(n.p, b.p, keep_bytes, hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) {
    n.release();
    return e;
  }
  b.release();
  b = n;
  return hipSuccess;
}
int grow_pool(vbx_ctx* ctx) {
  MapDev& m = ctx->map;
  hipStream_t s = ctx->stream;
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipMemcpy(&ctx->h_state, ctx->d_state, sizeof(DevState), hipMemcpyDeviceToHost));
  const uint64_t hard = ((1ull << 32) - 1) / m.nvox;  // voxel ids are 32 bit
  uint64_t limit = ctx->pool_limit ? std::min<uint64_t>(ctx->pool_limit, hard) : hard;
  const uint64_t new_cap = std::min<uint64_t>(2ull * m.cap_blocks, limit);
  auto give_up = [&]() {  // leave a consistent map behind: the keys of the failed pass go, pool_used is what exists
    const uint32_t used_now = std::min(ctx->h_state.pool_used, m.cap_blocks);
    (void)hipMemcpyAsync(&ctx->d_state->pool_used, &used_now, 4, hipMemcpyHostToDevice, s);
    (void)hipMemsetAsync(&ctx->d_state->new_count, 0, 4, s);
    (void)hipMemsetAsync(&ctx->d_state->error, 0, 4, s);
    hipLaunchKernelGGL(k_drop_unassigned_keys, grid_for((size_t)m.hmask + 1), dim3(256), 0, s, m, ctx->d_state);
    (void)hipStreamSynchronize(s);
  };
  if (new_cap <= m.cap_blocks) {
    give_up();
    ctx->fail("block pool / hash map capacity exceeded (max_blocks=%u%s)", m.cap_blocks,
              ctx->pool_limit ? ", growth limited by vbx_set_pool_limit" : ", 2^32 voxel ids");
    return VBX_ERR_CAPACITY;
  }
  const uint32_t used = std::min(ctx->h_state.pool_used, m.cap_blocks);
  const size_t nv_old = (size_t)used * m.nvox, nv_new = (size_t)new_cap * m.nvox;
  uint32_t hcap = 1;
  while (hcap < 4u * new_cap) hcap <<= 1;
  auto ok = [&](hipError_t e) { return e == hipSuccess; };
  bool good = ok(grow_buf(ctx->b_dist, nv_new * 4, nv_old * 4, 0, s)) && ok(grow_buf(ctx->b_weight, nv_new * 4, nv_old * 4, 0, s)) &&
              ok(grow_buf(ctx->b_rgba, nv_new * 4, nv_old * 4, 0, s)) &&
              ok(grow_buf(ctx->b_blkidx, (size_t)new_cap * 12, (size_t)used * 12, 0, s)) &&
              ok(grow_buf(ctx->b_blkflags, (size_t)new_cap * 4, (size_t)used * 4, 0, s)) &&
              ok(grow_buf(ctx->b_blkfirst, (size_t)new_cap * 8, (size_t)used * 8, 0xFF, s)) &&
              ok(grow_buf(ctx->b_freelist, (size_t)new_cap * 4, (size_t)std::min(ctx->h_state.free_count, m.cap_blocks) * 4, 0, s)) &&
              ok(grow_buf(ctx->b_newlist, (size_t)new_cap * 4, 0, 0, s)) &&
              ok(grow_buf(ctx->b_hkeys, (size_t)hcap * 8, 0, 0xFF, s)) && ok(grow_buf(ctx->b_hvals, (size_t)hcap * 4, 0, 0xFF, s));
  if (good && ctx->esdf_init)
    good = ok(grow_buf(ctx->b_edist, nv_new * 4, nv_old * 4, 0, s)) && ok(grow_buf(ctx->b_estate, nv_new * 4, nv_old * 4, 0, s)) &&
           ok(grow_buf(ctx->b_eraised, nv_new, nv_old, 0, s)) && ok(grow_buf(ctx->b_eactive, (size_t)new_cap * 4, (size_t)used * 4, 0, s));
  if (good && ctx->b_obs.p) good = ok(grow_buf(ctx->b_obs, nv_new * 4, nv_old * 4, 0, s));
  (void)hipGetLastError();
  if (!good) {
    give_up();
    ctx->fail("block pool full (max_blocks=%u) and no device memory to grow it to %llu blocks", m.cap_blocks,
              (unsigned long long)new_cap);
    return VBX_ERR_CAPACITY;
  }
  // per-voxel claim arrays of the Fast solver are sized by the pool: start over (vbx_host_tsdf.hpp `fresh`)
  ctx->b_own0.release();
  ctx->b_own1.release();
  ctx->b_cl.release();
  m.cap_blocks = (uint32_t)new_cap;
  ctx->hcap = hcap;
  m.hmask = hcap - 1;
  m.hkeys = ctx->b_hkeys.as<uint64_t>();
  m.hvals = ctx->b_hvals.as<uint32_t>();
  m.dist = ctx->b_dist.as<float>();
  m.weight = ctx->b_weight.as<float>();
  m.rgba = ctx->b_rgba.as<uint32_t>();
  m.blk_idx = ctx->b_blkidx.as<int32_t>();
  m.blk_flags = ctx->b_blkflags.as<uint32_t>();
  m.blk_first = ctx->track_block_order ? ctx->b_blkfirst.as<unsigned long long>() : nullptr;
  m.free_list = ctx->b_freelist.as<uint32_t>();
  // pool_used may have been pushed to the old capacity by the failed commit: the blocks that exist are `used`
  HIP_TRY(hipMemcpyAsync(&ctx->d_state->pool_used, &used, 4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(&ctx->d_state->new_count, 0, 4, s));
  HIP_TRY(hipMemsetAsync(&ctx->d_state->error, 0, 4, s));
  if (used) KLAUNCH(k_rehash, grid_for(used), dim3(256), 0, s, m, ctx->d_state);
  HIP_TRY(hipStreamSynchronize(s));
  ++ctx->pool_grown;
  return VBX_OK;
}

RayTab make_tab_c(vbx_ctx* ctx, uint32_t R) {  // table C
  RayTab t;
  t.px = ctx->w_px.as<float>(); t.py = ctx->w_py.as<float>(); t.pz = ctx->w_pz.as<float>();
  t.rgba = ctx->w_rgba.as<uint32_t>(); t.w = ctx->w_w.as<float>();
  t.flags = ctx->w_flags.as<uint8_t>(); t.bkey = ctx->w_bkey.as<uint64_t>();
  t.R = R;
  return t;
}
int ensure_tab_c(vbx_ctx* ctx, size_t R) {
  const size_t n = R + 1;
  HIP_TRY(ctx->w_px.ensure(n * 4)); HIP_TRY(ctx->w_py.ensure(n * 4)); HIP_TRY(ctx->w_pz.ensure(n * 4));
  HIP_TRY(ctx->w_rgba.ensure(n * 4)); HIP_TRY(ctx->w_w.ensure(n * 4)); HIP_TRY(ctx->w_flags.ensure(n));
  HIP_TRY(ctx->w_bkey.ensure(n * 8));
  return VBX_OK;
}

RayTab make_tab(vbx_ctx* ctx, bool second, uint32_t R) {
  RayTab t;
  if (!second) {
    t.px = ctx->t_px.as<float>(); t.py = ctx->t_py.as<float>(); t.pz = ctx->t_pz.as<float>();
    t.rgba = ctx->t_rgba.as<uint32_t>(); t.w = ctx->t_w.as<float>();
    t.flags = ctx->t_flags.as<uint8_t>(); t.bkey = ctx->t_bkey.as<uint64_t>();
  } else {
    t.px = ctx->u_px.as<float>(); t.py = ctx->u_py.as<float>(); t.pz = ctx->u_pz.as<float>();
    t.rgba = ctx->u_rgba.as<uint32_t>(); t.w = ctx->u_w.as<float>();
    t.flags = ctx->u_flags.as<uint8_t>(); t.bkey = ctx->u_bkey.as<uint64_t>();
  }
  t.R = R;
  return t;
}

int ensure_tab(vbx_ctx* ctx, bool second, size_t R, bool with_bkey) {
  const size_t n = R + 1;
  if (!second) {
    HIP_TRY(ctx->t_px.ensure(n * 4)); HIP_TRY(ctx->t_py.ensure(n * 4)); HIP_TRY(ctx->t_pz.ensure(n * 4));
    HIP_TRY(ctx->t_rgba.ensure(n * 4)); HIP_TRY(ctx->t_w.ensure(n * 4)); HIP_TRY(ctx->t_flags.ensure(n));
    if (with_bkey) HIP_TRY(ctx->t_bkey.ensure(n * 8));
  } else {
    HIP_TRY(ctx->u_px.ensure(n * 4)); HIP_TRY(ctx->u_py.ensure(n * 4)); HIP_TRY(ctx->u_pz.ensure(n * 4));
    HIP_TRY(ctx->u_rgba.ensure(n * 4)); HIP_TRY(ctx->u_w.ensure(n * 4)); HIP_TRY(ctx->u_flags.ensure(n));
    if (with_bkey) HIP_TRY(ctx->u_bkey.ensure(n * 8));
  }
  return VBX_OK;
}

// Exclusive prefix sum, one launch (k_scan_excl, vbx_sort.hpp).
int exclusive_scan_u32(vbx_ctx* ctx, uint32_t* in, uint32_t* out, size_t n) {
  if (n == 0) return VBX_OK;
  if (n > 0xFFFFF000ull) {
    ctx->fail("exclusive scan: too many elements");
    return VBX_ERR_INVALID;
  }
  const uint32_t ntiles = (uint32_t)((n + kScanTile - 1) / kScanTile);
  if (ctx->b_scan_desc.cap < (size_t)(ntiles + 1) * 8 + 8) {
    // descriptors + the ticket counter; zero = "no word of any generation"
    HIP_TRY(ctx->b_scan_desc.ensure(std::max<size_t>((size_t)(ntiles + 1) * 8 + 8, 1 << 16)));
    HIP_TRY(hipMemsetAsync(ctx->b_scan_desc.p, 0, ctx->b_scan_desc.cap, ctx->stream));
    ctx->scan_ticket_base = 0;
    ctx->scan_gen = 0;
  }
  if (++ctx->scan_gen >= 0x3FFFFFFFu) {  // generation tags about to repeat: start over
    HIP_TRY(hipMemsetAsync(ctx->b_scan_desc.p, 0, ctx->b_scan_desc.cap, ctx->stream));
    ctx->scan_ticket_base = 0;
    ctx->scan_gen = 1;
  }
  unsigned long long* desc = ctx->b_scan_desc.as<unsigned long long>() + 1;
  uint32_t* ticket = ctx->b_scan_desc.as<uint32_t>();
  KLAUNCH(k_scan_excl, dim3(ntiles), dim3(kScanThreads), 0, ctx->stream, in, out, (uint32_t)n, desc, ticket,
          ctx->scan_ticket_base, ctx->scan_gen);
  ctx->scan_ticket_base += ntiles;  // wraps like the device counter
  return VBX_OK;
}

// Stable sort of ctx->b_keys0 (and b_vals0 when with_vals) on key bits [begin_bit, end_bit); the
// result is left in b_keys1 / b_vals1 (the DBuf handles are swapped when an even number of
// passes ends in the input buffers, so callers must fetch the pointers after the call).
// See vbx_sort.hpp.
template <int CAP>
int rsort_pass(vbx_ctx* ctx, const uint64_t* kin, const uint32_t* vin, uint64_t* kout, uint32_t* vout, uint32_t n,
               const uint32_t* n_dev, int shift, uint32_t nwg, bool with_vals) {
  hipStream_t s = ctx->stream;
  uint32_t* hist = ctx->b_hist0.as<uint32_t>();
  uint32_t* gofs = ctx->b_hist1.as<uint32_t>();
  KLAUNCH(k_rsort_count<CAP>, dim3(nwg), dim3(kSortThreads), 0, s, kin, n, n_dev, shift, hist, nwg);
  int rc = exclusive_scan_u32(ctx, hist, gofs, (size_t)(1u << CAP) * nwg);
  if (rc) return rc;
  if (with_vals)
    KLAUNCH((k_rsort_scatter<CAP, true>), dim3(nwg), dim3(kSortThreads), 0, s, kin, vin, kout, vout, n, n_dev, shift,
                       gofs, nwg);
  else
    KLAUNCH((k_rsort_scatter<CAP, false>), dim3(nwg), dim3(kSortThreads), 0, s, kin, vin, kout, vout, n, n_dev,
                       shift, gofs, nwg);
  return VBX_OK;
}
// The fused variant of a sort (vbx_sort.hpp): one histogram launch, then one launch per pass.
constexpr int kFsRing = 256;  // histogram slots between two clears
template <int BITS>
int fused_pass(vbx_ctx* ctx, const uint64_t* kin, const uint32_t* vin, uint64_t* kout, uint32_t* vout, uint32_t n,
               const uint32_t* n_dev, int shift, uint32_t ntiles, bool with_vals, const uint32_t* hist) {
  if (++ctx->fs_gen >= 0x7FFFFFFFu) {  // generation tags about to repeat: start over
    HIP_TRY(hipMemsetAsync(ctx->b_fs_desc.p, 0, ctx->b_fs_desc.cap, ctx->stream));
    ctx->fs_ticket_base = 0;
    ctx->fs_gen = 1;
  }
  // ticket | tile flags | group flags | group totals | tile count rows (laid out for fs_desc_tiles tiles)
  const size_t T = ctx->fs_desc_tiles, G = T / kFsGroupBig;
  const uint32_t gsize = ntiles <= (uint32_t)kFsGroup ? (uint32_t)kFsGroup : (uint32_t)kFsGroupBig;
  uint32_t* ticket = ctx->b_fs_desc.as<uint32_t>();
  unsigned long long* flags = ctx->b_fs_desc.as<unsigned long long>() + 1;
  unsigned long long* gflags = flags + T;
  uint32_t* gpre = reinterpret_cast<uint32_t*>(gflags + G);
  uint16_t* cnt16 = reinterpret_cast<uint16_t*>(gpre + G * (1 << kFsMaxBits));
  if (with_vals)
    KLAUNCH((k_rsort_fused<BITS, true>), dim3(ntiles), dim3(kFsThreads), 0, ctx->stream, kin, vin, kout, vout, n, n_dev,
            shift, hist, cnt16, flags, ticket, ctx->fs_ticket_base, ctx->fs_gen, ctx->d_state, gflags, gpre, gsize);
  else
    KLAUNCH((k_rsort_fused<BITS, false>), dim3(ntiles), dim3(kFsThreads), 0, ctx->stream, kin, vin, kout, vout, n,
            n_dev, shift, hist, cnt16, flags, ticket, ctx->fs_ticket_base, ctx->fs_gen, ctx->d_state, gflags, gpre, gsize);
  ctx->fs_ticket_base += ntiles;  // wraps like the device counter
  return VBX_OK;
}
int stable_sort_fused(vbx_ctx* ctx, uint32_t n, unsigned begin_bit, unsigned end_bit, bool with_vals,
                      const uint32_t* n_dev) {
  const unsigned bits = end_bit - begin_bit;
  FsPasses ps{};
  ps.np = (int)((bits + kFsMaxBits - 1) / kFsMaxBits);
  const unsigned per = (bits + ps.np - 1) / ps.np;
  for (int p = 0; p < ps.np; ++p) {
    const unsigned sh = begin_bit + p * per, w = std::min(per, end_bit - sh);
    ps.shift[p] = (int)sh;
    ps.mask[p] = (1u << w) - 1u;
  }
  const uint32_t ntiles = (n + kFsTile - 1) / kFsTile;
  const size_t slot_bytes = (size_t)kFsMaxPasses * (1 << kFsMaxBits) * 4;
  if (ctx->b_fs_hist.cap < slot_bytes * kFsRing) {
    HIP_TRY(ctx->b_fs_hist.ensure(slot_bytes * kFsRing));
    HIP_TRY(hipMemsetAsync(ctx->b_fs_hist.p, 0, ctx->b_fs_hist.cap, ctx->stream));
    ctx->fs_ring_pos = 0;
  }
  if (ntiles > ctx->fs_desc_tiles) {  // descriptors laid out for a whole number of groups; zero = "no word of any generation"
    const size_t T = ((size_t)ntiles + kFsGroup - 1) / kFsGroup * kFsGroup, G = T / kFsGroupBig;
    const size_t desc_bytes = 8 + T * 8 + G * 8 + G * (1 << kFsMaxBits) * 4 + T * (1 << kFsMaxBits) * 2;
    HIP_TRY(ctx->b_fs_desc.ensure(desc_bytes));
    HIP_TRY(hipMemsetAsync(ctx->b_fs_desc.p, 0, ctx->b_fs_desc.cap, ctx->stream));
    ctx->fs_desc_tiles = (uint32_t)T;
    ctx->fs_ticket_base = 0;
    ctx->fs_gen = 0;
  }
  if (ctx->fs_ring_pos == kFsRing) {  // every slot used once: clear them all (stream-ordered behind their readers)
    HIP_TRY(hipMemsetAsync(ctx->b_fs_hist.p, 0, slot_bytes * kFsRing, ctx->stream));
    ctx->fs_ring_pos = 0;
  }
  uint32_t* hist = ctx->b_fs_hist.as<uint32_t>() + (size_t)ctx->fs_ring_pos++ * kFsMaxPasses * (1 << kFsMaxBits);
  HIP_TRY(ctx->b_keys1.ensure((size_t)n * 8));
  if (with_vals) HIP_TRY(ctx->b_vals1.ensure((size_t)n * 4));
  KLAUNCH(k_rsort_hist, dim3(std::min<uint32_t>(ntiles, 512u)), dim3(kFsThreads), 0, ctx->stream, ctx->b_keys0.as<uint64_t>(), n, n_dev, ps, hist);
  bool in0 = true;
  for (int p = 0; p < ps.np; ++p) {
    const uint64_t* kin = (in0 ? ctx->b_keys0 : ctx->b_keys1).as<uint64_t>();
    uint64_t* kout = (in0 ? ctx->b_keys1 : ctx->b_keys0).as<uint64_t>();
    const uint32_t* vin = with_vals ? (in0 ? ctx->b_vals0 : ctx->b_vals1).as<uint32_t>() : nullptr;
    uint32_t* vout = with_vals ? (in0 ? ctx->b_vals1 : ctx->b_vals0).as<uint32_t>() : nullptr;
    const uint32_t* h = hist + (size_t)p * (1 << kFsMaxBits);
    int w = 0;
    while ((1u << w) <= ps.mask[p]) ++w;
    int rc;
    switch (w) {
      case 4: rc = fused_pass<4>(ctx, kin, vin, kout, vout, n, n_dev, ps.shift[p], ntiles, with_vals, h); break;
      case 5: rc = fused_pass<5>(ctx, kin, vin, kout, vout, n, n_dev, ps.shift[p], ntiles, with_vals, h); break;
      case 6: rc = fused_pass<6>(ctx, kin, vin, kout, vout, n, n_dev, ps.shift[p], ntiles, with_vals, h); break;
      case 7: rc = fused_pass<7>(ctx, kin, vin, kout, vout, n, n_dev, ps.shift[p], ntiles, with_vals, h); break;
      case 8: rc = fused_pass<8>(ctx, kin, vin, kout, vout, n, n_dev, ps.shift[p], ntiles, with_vals, h); break;
      case 9: rc = fused_pass<9>(ctx, kin, vin, kout, vout, n, n_dev, ps.shift[p], ntiles, with_vals, h); break;
      default: rc = fused_pass<10>(ctx, kin, vin, kout, vout, n, n_dev, ps.shift[p], ntiles, with_vals, h); break;
    }
    if (rc) return rc;
    in0 = !in0;
  }
  if (in0) {
    std::swap(ctx->b_keys0, ctx->b_keys1);
    if (with_vals) std::swap(ctx->b_vals0, ctx->b_vals1);
  }
  return VBX_OK;
}

// n_dev (optional): the number of keys lives on the device; n64 is then the host's upper bound (grid, buffers).
int stable_sort01(vbx_ctx* ctx, size_t n64, unsigned begin_bit, unsigned end_bit, bool with_vals,
                  const uint32_t* n_dev = nullptr) {
  if (n64 == 0 || end_bit <= begin_bit) {
    // nothing to order: the "sorted" data is the input
    std::swap(ctx->b_keys0, ctx->b_keys1);
    if (with_vals) std::swap(ctx->b_vals0, ctx->b_vals1);
    return VBX_OK;
  }
  if (n64 > 0xFFFFF000ull) {
    ctx->fail("stable sort: too many keys");
    return VBX_ERR_INVALID;
  }
  const uint32_t n = (uint32_t)n64;
  const unsigned bits = end_bit - begin_bit;
  if (ctx->fs_enabled < 0) {
    const char* e = getenv("VBX_SORT_FUSED");  // =0 selects the three-launch passes
    ctx->fs_enabled = (e && e[0] == '0') ? 0 : 1;
  }
  if (ctx->fs_enabled && bits <= (unsigned)kFsMaxBits * kFsMaxPasses &&
      (bits + ((bits + kFsMaxBits - 1) / kFsMaxBits) - 1) / ((bits + kFsMaxBits - 1) / kFsMaxBits) >= 4)
    return stable_sort_fused(ctx, n, begin_bit, end_bit, with_vals, n_dev);
  const unsigned passes = (bits + kSortMaxBits - 1) / kSortMaxBits;
  const unsigned per = (bits + passes - 1) / passes;  // digit width, <= 12
  const uint32_t nwg = (n + kSortTile - 1) / kSortTile;
  // the kernels are instantiated for every digit width 1..12, so a pass never looks at bits
  // outside [begin_bit, end_bit)
  HIP_TRY(ctx->b_hist0.ensure(((size_t)1 << kSortMaxBits) * nwg * 4));
  HIP_TRY(ctx->b_hist1.ensure(((size_t)1 << kSortMaxBits) * nwg * 4));
  HIP_TRY(ctx->b_keys1.ensure((size_t)n * 8));
  if (with_vals) HIP_TRY(ctx->b_vals1.ensure((size_t)n * 4));
  bool in0 = true;  // where the current input lives
  unsigned shift = begin_bit;
  while (shift < end_bit) {
    const unsigned w = std::min(per, end_bit - shift);
    const uint64_t* kin = (in0 ? ctx->b_keys0 : ctx->b_keys1).as<uint64_t>();
    uint64_t* kout = (in0 ? ctx->b_keys1 : ctx->b_keys0).as<uint64_t>();
    const uint32_t* vin = with_vals ? (in0 ? ctx->b_vals0 : ctx->b_vals1).as<uint32_t>() : nullptr;
    uint32_t* vout = with_vals ? (in0 ? ctx->b_vals1 : ctx->b_vals0).as<uint32_t>() : nullptr;
    int rc;
    switch (w) {
      case 1: rc = rsort_pass<1>(ctx, kin, vin, kout, vout, n, n_dev, (int)shift, nwg, with_vals); break;
      case 2: rc = rsort_pass<2>(ctx, kin, vin, kout, vout, n, n_dev, (int)shift, nwg, with_vals); break;
      case 3: rc = rsort_pass<3>(ctx, kin, vin, kout, vout, n, n_dev, (int)shift, nwg, with_vals); break;
      case 4: rc = rsort_pass<4>(ctx, kin, vin, kout, vout, n, n_dev, (int)shift, nwg, with_vals); break;
      case 5: rc = rsort_pass<5>(ctx, kin, vin, kout, vout, n, n_dev, (int)shift, nwg, with_vals); break;
      case 6: rc = rsort_pass<6>(ctx, kin, vin, kout, vout, n, n_dev, (int)shift, nwg, with_vals); break;
      case 7: rc = rsort_pass<7>(ctx, kin, vin, kout, vout, n, n_dev, (int)shift, nwg, with_vals); break;
      case 8: rc = rsort_pass<8>(ctx, kin, vin, kout, vout, n, n_dev, (int)shift, nwg, with_vals); break;
      case 9: rc = rsort_pass<9>(ctx, kin, vin, kout, vout, n, n_dev, (int)shift, nwg, with_vals); break;
      case 10: rc = rsort_pass<10>(ctx, kin, vin, kout, vout, n, n_dev, (int)shift, nwg, with_vals); break;
      case 11: rc = rsort_pass<11>(ctx, kin, vin, kout, vout, n, n_dev, (int)shift, nwg, with_vals); break;
      default: rc = rsort_pass<12>(ctx, kin, vin, kout, vout, n, n_dev, (int)shift, nwg, with_vals); break;
    }
    if (rc) return rc;
    in0 = !in0;
    shift += w;
  }
  if (in0) {  // an even number of passes: the result is in the "0" buffers
    std::swap(ctx->b_keys0, ctx->b_keys1);
    if (with_vals) std::swap(ctx->b_vals0, ctx->b_vals1);
  }
  return VBX_OK;
}

inline unsigned bits_for(uint64_t v) {
  unsigned b = 1;
  while (b < 64 && (v >> b)) ++b;
  return b;
}

void tmark(vbx_ctx* ctx, int i) {
  if (ctx->timing) {
    (void)hipEventRecord(ctx->ev[i], ctx->stream);
    ctx->ev_hit[i] = true;
  }
}

CastCfg make_cast_cfg(vbx_ctx* ctx, const vbx_tsdf_cfg* cfg, const float pos[3]) {
  CastCfg c{};
  c.origin = {pos[0], pos[1], pos[2]};
  c.trunc = cfg->default_truncation_distance;
  c.max_ray_length_m = cfg->max_ray_length_m;
  c.min_ray_length_m = cfg->min_ray_length_m;
  c.max_weight = cfg->max_weight;
  c.sparsity_factor = cfg->sparsity_compensation_factor;
  c.carving = cfg->voxel_carving_enabled != 0;
  // tsdf_integrator.cc:62-65: clearing rays have no utility if voxel_carving is disabled
  c.allow_clear = (cfg->allow_clear != 0) && (cfg->voxel_carving_enabled != 0);
  c.use_const_weight = cfg->use_const_weight != 0;
  c.dropoff = cfg->use_weight_dropoff != 0;
  c.sparsity = cfg->use_sparsity_compensation_factor != 0;
  c.anti_grazing = cfg->enable_anti_grazing != 0;
  c.max_consecutive = cfg->max_consecutive_ray_collisions;
  c.start_factor_times_inv = cfg->start_voxel_subsampling_factor * ctx->map.voxel_size_inv;
  c.take_limit = ~0u;
  return c;
}


}  // namespace

