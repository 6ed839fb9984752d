// vbx_host_esdf.hpp — host orchestration of the ESDF update and addNewRobotPosition
// Part of libvbx_hip.so's single translation unit (included by vbx_hip.hip, in order).

namespace {
__global__ void k_esdf_reset_flags(MapDev m, EsdfDev e, uint32_t n_slots, int drop_layer, int keep_classify_pending,
                                   DevState* st) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s == 0) {  // the update's counters
    st->esdf_blocks = 0;
    st->esdf_raise_any = 0;
    st->esdf_relax_blocks = 0;
    st->esdf_phase_changed[0] = 0;
    st->esdf_phase_changed[1] = 0;
  }
  if (s >= n_slots) return;
  // blocks addNewRobotPosition left work in: 8 = take part in this update (their voxels sit in
  // open_/raise_), 16 = also re-run the TSDF classification on them
  const uint32_t f = m.blk_flags[s];
  const uint32_t pend = f & (kFlagEsdfPendClassify | kFlagEsdfPendOpen);
  e.active[s] = ((pend && !drop_layer) ? (8u | ((pend & kFlagEsdfPendClassify) ? 16u : 0u)) : 0u) |
                (((f & kFlagEsdfUnsettled) && !drop_layer) ? 64u : 0u);  // uploaded voxels: a full first pass when the block runs
  // updateFromTsdfBlocks does not consume updated_blocks_: the classification stays pending
  uint32_t nf = f & ~((keep_classify_pending ? 0u : kFlagEsdfPendClassify) | kFlagEsdfPendOpen);
  if (drop_layer) nf &= ~(kFlagEsdfAlloc | (kFlagUpdMask << kFlagEsdfUpdShift) | kFlagEsdfUnsettled);
  if (nf != f) m.blk_flags[s] = nf;
}
// End of a speculatively queued update (esdf_update_t): drops the raise marks of the touched blocks and, on request, the
// TSDF blocks' Update::kEsdf bit — unless a phase did not converge in the sweeps queued for it (the host then finishes
// the update sweep by sweep and clears with the plain calls).
__global__ void k_esdf_finish(MapDev m, EsdfDev e, uint32_t nvox, DevState* st, TileGuard gd, int clear_tsdf_bit) {
  if (gd.guard0 && st->esdf_phase_changed[0] >= gd.guard0) return;
  if (gd.guard1 && st->esdf_phase_changed[1] >= gd.guard1) return;
  const uint32_t slot = blockIdx.x;
  const uint32_t a = e.active[slot];
  if (clear_tsdf_bit && threadIdx.x == 0 && (a & 8u)) m.blk_flags[slot] &= ~4u;  // updated().reset(Update::kEsdf), esdf_integrator.cc:113-121
  if (!(a & 4u) || !(gd.force || st->esdf_raise_any)) return;
  uint32_t* r = reinterpret_cast<uint32_t*>(e.raised + (size_t)slot * nvox);
  for (uint32_t i = threadIdx.x; i < nvox / 4; i += blockDim.x) r[i] = 0;
}
__global__ void k_esdf_clear_tsdf_bit(MapDev m, EsdfDev e, uint32_t n_slots) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  if (e.active[s] & 8u) m.blk_flags[s] &= ~4u;  // updated().reset(Update::kEsdf), esdf_integrator.cc:113-121
}

EsdfDev esdf_dev(vbx_ctx* ctx) {
  EsdfDev e;
  e.dist = ctx->b_edist.as<float>();
  e.state = ctx->b_estate.as<uint32_t>();
  e.raised = ctx->b_eraised.as<uint8_t>();
  e.active = ctx->b_eactive.as<uint32_t>();
  return e;
}

int esdf_ensure(vbx_ctx* ctx) {
  if (ctx->esdf_init) return VBX_OK;
  const MapDev& m = ctx->map;
  const size_t nv = (size_t)m.cap_blocks * m.nvox;
  HIP_TRY(ctx->b_edist.ensure(nv * 4));
  HIP_TRY(ctx->b_estate.ensure(nv * 4));
  HIP_TRY(ctx->b_eraised.ensure(nv));
  HIP_TRY(ctx->b_eactive.ensure((size_t)m.cap_blocks * 4));
  HIP_TRY(hipMemsetAsync(ctx->b_edist.p, 0, nv * 4, ctx->stream));
  HIP_TRY(hipMemsetAsync(ctx->b_estate.p, 0, nv * 4, ctx->stream));
  HIP_TRY(hipMemsetAsync(ctx->b_eraised.p, 0, nv, ctx->stream));
  HIP_TRY(hipMemsetAsync(ctx->b_eactive.p, 0, (size_t)m.cap_blocks * 4, ctx->stream));
  ctx->esdf_init = true;
  return VBX_OK;
}

template <int VPS, bool FULL>
int esdf_phase(vbx_ctx* ctx, const EsdfDev& e, const EsdfCfgDev& c, int mode, uint32_t used,
               uint32_t* sweeps, uint32_t* g_sweep) {
  hipStream_t s = ctx->stream;
  HIP_TRY(hipMemsetAsync(&ctx->d_state->changed, 0, 4, s));
  if (FULL && mode == 1) {
    // one block colour per launch (see k_esdf_tile); a round of all eight colours without a
    // change is the fixed point
    uint32_t sub = 0;
    for (;;) {
      for (int i = 0; i < 8; ++i) {
        ++sub;
        KLAUNCH((k_esdf_tile<VPS, true>), dim3(used), dim3(kEsdfThreads), 0, s, ctx->map, e, c, 1, sub,
                           ctx->d_state);
        KLAUNCH(k_esdf_rotate_active_colour, grid_for(used), dim3(256), 0, s, ctx->map, e, used,
                           (int)(sub & 7u));
      }
      int rc = sync_state(ctx);
      if (rc) return rc;
      *sweeps += 1;
      if (ctx->h_state.changed + 8 <= sub) return VBX_OK;
      if (sub > 800000) {
        ctx->fail("ESDF: wavefront did not converge");
        return VBX_ERR_HIP;
      }
    }
  }
  if (mode == 2) {
    ++*g_sweep;
    KLAUNCH((k_esdf_tile<VPS, FULL>), dim3(used), dim3(kEsdfThreads), 0, s, ctx->map, e, c, mode, 1u, ctx->d_state,
                       FULL ? 0u : *g_sweep, 1);
    ++*sweeps;
    return VBX_OK;
  }
  // Global sweeps until none changes a block.  The active-block flags live on the device, so
  // several sweeps are queued per host check (an idle sweep is ~10 us: every workgroup leaves at
  // once; a check costs a read-back); the kernels record the last sweep that changed anything.
  uint32_t sweep_no = 0;
  for (;;) {
    constexpr int kPerCheck = 3;
    for (int i = 0; i < kPerCheck; ++i) {
      ++sweep_no;
      ++*g_sweep;
      // quasi-Euclidean: the blocks of a sweep are found by their tag (= update-wide sweep number; the first sweep of a
      // phase takes everything touched so far) — no rotate launch between the sweeps
      KLAUNCH((k_esdf_tile<VPS, FULL>), dim3(used), dim3(kEsdfThreads), 0, s, ctx->map, e, c, mode, sweep_no,
                         ctx->d_state, FULL ? 0u : *g_sweep, sweep_no == 1 ? 1 : 0);
      if (FULL) KLAUNCH(k_esdf_rotate_active, grid_for(used), dim3(256), 0, s, e, used, 0);
    }
    int rc = sync_state(ctx);
    if (rc) return rc;
    if (ctx->h_state.changed < sweep_no) {  // the last queued sweep was idle: fixed point
      *sweeps += std::max<uint32_t>(ctx->h_state.changed + 1, sweep_no - kPerCheck + 1) - (sweep_no - kPerCheck);
      return VBX_OK;
    }
    *sweeps += kPerCheck;
    if (sweep_no > 100000) {
      ctx->fail("ESDF: wavefront did not converge");
      return VBX_ERR_HIP;
    }
  }
}

template <int VPS, bool FULL>
int esdf_update_t(vbx_ctx* ctx, const vbx_esdf_cfg* cfg, int batch, int clear_updated_flag,
                  const int32_t* list = nullptr, size_t n_list = 0, int list_incremental = 0) {
  MapDev& m = ctx->map;
  hipStream_t s = ctx->stream;
  int rc = esdf_ensure(ctx);
  if (rc) return rc;
  rc = sync_state(ctx);
  if (rc) return rc;
  const uint32_t used = ctx->h_state.pool_used;
  ctx->counters = vbx_counters{};
  if (used == 0) return VBX_OK;
  EsdfDev e = esdf_dev(ctx);
  EsdfCfgDev c;
  c.max_distance = cfg->max_distance_m;
  c.min_distance = cfg->min_distance_m;
  c.default_distance = cfg->default_distance_m;
  c.min_diff = cfg->min_diff_m;
  c.min_weight = cfg->min_weight;
  c.add_occupied_crust = cfg->add_occupied_crust != 0;
  c.voxel_size = m.voxel_size;
  const size_t nv = (size_t)used * m.nvox;
  for (int i = 0; i < 9; ++i) ctx->ev_hit[i] = false;
  tmark(ctx, 0);
  if (batch) {  // esdf_layer_->removeAllBlocks(), esdf_integrator.cc:95
    HIP_TRY(hipMemsetAsync(e.dist, 0, nv * 4, s));
    HIP_TRY(hipMemsetAsync(e.state, 0, nv * 4, s));
  }
  // `raised` is clear between updates except for the marks addNewRobotPosition left (a batch
  // update drops those with the layer)
  if (batch) HIP_TRY(hipMemsetAsync(e.raised, 0, nv, s));
  const bool robot_pending = ctx->esdf_robot_pending && !batch;
  ctx->esdf_robot_pending = false;
  KLAUNCH(k_esdf_reset_flags, grid_for(used), dim3(256), 0, s, m, e, used, batch ? 1 : 0, list ? 1 : 0, ctx->d_state);
  if (list) {
    HIP_TRY(ctx->b_head.ensure(std::max<size_t>(n_list, 1) * 12));
    HIP_TRY(hipMemcpyAsync(ctx->b_head.p, list, n_list * 12, hipMemcpyHostToDevice, s));
    if (n_list)
      KLAUNCH(k_esdf_mark_listed, grid_for(n_list), dim3(256), 0, s, m, e, ctx->b_head.as<int32_t>(),
                         (uint32_t)n_list);
    KLAUNCH(k_esdf_classify, dim3(used, (m.nvox + 255) / 256), dim3(256), 0, s, m, e, c,
                       list_incremental ? 1 : 0, 2, ctx->d_state);
  } else {
    KLAUNCH(k_esdf_classify, dim3(used, (m.nvox + 255) / 256), dim3(256), 0, s, m, e, c,
                       batch ? 0 : 1, batch ? 0 : 1, ctx->d_state);
  }
  KLAUNCH(k_esdf_seed_active, grid_for((size_t)used * 27), dim3(256), 0, s, m, e, used);
  uint32_t sweeps = 0;
  uint32_t g_sweep = 0;  // sweep number across the phases of this update (the tile kernel's scheduling tag)
  // The wavefronts run to their exact fixed points: min_diff_m only gates the TSDF->ESDF copy
  // of phase 1 (see HISTORY.md §4.4 for why the relaxation itself uses strict improvement).
  EsdfCfgDev cr = c;
  cr.min_diff = 0.0f;
  if (!FULL && (m.nvox & 3u) == 0) {
    // Quasi-Euclidean: the whole update is queued behind ONE read-back.  A sweep that finds nothing to do costs
    // ~4.5 us on the device, a host check ~13 us of idle stream plus the idle sweeps queued in front of it — and the
    // phases of a frame-to-frame update need 3-4 sweeps each (tools/esdf_timeline.sh).  Every launch behind the
    // raise sweeps carries the number of the last raise sweep, every launch behind the lower sweeps that of the last
    // lower sweep: if a phase was still changing blocks in its last queued sweep the later launches leave at once,
    // and the update is finished below the way it used to run (sweep by sweep, with a check every third).
    if (ctx->esdf_spec_raise <= 0) {
      const char* e1 = getenv("VBX_ESDF_RAISE_SWEEPS");
      const char* e2 = getenv("VBX_ESDF_LOWER_SWEEPS");
      ctx->esdf_spec_raise = e1 ? std::max(1, atoi(e1)) : 6;
      ctx->esdf_spec_lower = e2 ? std::max(1, atoi(e2)) : 6;
    }
    tmark(ctx, 1);
    TileGuard gd{0, 0, 1, robot_pending ? 1 : 0};
    for (int i = 0; i < ctx->esdf_spec_raise; ++i) {
      ++g_sweep;
      KLAUNCH((k_esdf_tile<VPS, FULL>), dim3(used), dim3(kEsdfThreads), 0, s, m, e, cr, 0, g_sweep, ctx->d_state, g_sweep,
              i == 0 ? 1 : 0, gd);
    }
    const uint32_t g_raise = g_sweep;
    tmark(ctx, 3);
    gd.guard0 = g_raise;
    for (int i = 0; i < ctx->esdf_spec_lower; ++i) {
      ++g_sweep;
      KLAUNCH((k_esdf_tile<VPS, FULL>), dim3(used), dim3(kEsdfThreads), 0, s, m, e, cr, 1, g_sweep - g_raise, ctx->d_state,
              g_sweep, i == 0 ? 1 : 0, gd);
    }
    const uint32_t g_lower = g_sweep;
    gd.guard1 = g_lower;
    ++g_sweep;
    KLAUNCH((k_esdf_tile<VPS, FULL>), dim3(used), dim3(kEsdfThreads), 0, s, m, e, cr, 2, 1u, ctx->d_state, g_sweep, 1, gd);
    tmark(ctx, 6);
    KLAUNCH(k_esdf_finish, dim3(used), dim3(256), 0, s, m, e, m.nvox, ctx->d_state, gd,
            (clear_updated_flag && !batch) ? 1 : 0);
    tmark(ctx, 7);
    rc = sync_state(ctx);
    if (rc) return rc;
    ctx->counters.esdf_blocks = ctx->h_state.esdf_blocks;
    const bool any = ctx->h_state.esdf_blocks || robot_pending;
    const bool raise_any = any && (ctx->h_state.esdf_raise_any || robot_pending);
    const uint32_t ch0 = ctx->h_state.esdf_phase_changed[0], ch1 = ctx->h_state.esdf_phase_changed[1];
    const bool raise_done = !raise_any || ch0 < g_raise;
    const bool lower_done = !any || (raise_done && ch1 < g_lower);
    if (raise_any) sweeps += std::min(ch0 + 1, g_raise);
    if (any && raise_done) sweeps += (ch1 > g_raise ? std::min(ch1 - g_raise + 1, g_lower - g_raise) : 1) + 1;
    if (!lower_done) {
      ctx->counters.esdf_respeculated += 1;
      if (!raise_done) {
        rc = esdf_phase<VPS, FULL>(ctx, e, cr, 0, used, &sweeps, &g_sweep);
        if (rc) return rc;
      }
      rc = esdf_phase<VPS, FULL>(ctx, e, cr, 1, used, &sweeps, &g_sweep);
      if (rc) return rc;
      rc = esdf_phase<VPS, FULL>(ctx, e, cr, 2, used, &sweeps, &g_sweep);
      if (rc) return rc;
      if (raise_any) HIP_TRY(hipMemsetAsync(e.raised, 0, nv, s));
      if (clear_updated_flag && !batch) KLAUNCH(k_esdf_clear_tsdf_bit, grid_for(used), dim3(256), 0, s, m, e, used);
      tmark(ctx, 7);
      rc = sync_state(ctx);
      if (rc) return rc;
    }
    ctx->counters.esdf_sweeps = sweeps;
    ctx->counters.esdf_relaxations = ctx->h_state.esdf_relax_blocks;
    if (ctx->timing) {
      (void)hipEventSynchronize(ctx->ev[7]);
      vbx_timing& o = ctx->last_timing;
      o = vbx_timing{};
      float t = 0;
      (void)hipEventElapsedTime(&o.total_ms, ctx->ev[0], ctx->ev[7]);
      (void)hipEventElapsedTime(&t, ctx->ev[0], ctx->ev[1]);
      o.prep_ms = t;  // phase 1 (classification)
      (void)hipEventElapsedTime(&t, ctx->ev[1], ctx->ev[3]); o.solve_ms = t;  // raise
      (void)hipEventElapsedTime(&t, ctx->ev[3], ctx->ev[6]); o.fold_ms = t;   // lower + canonical parents
    }
    return VBX_OK;
  }
  rc = sync_state(ctx);
  if (rc) return rc;
  tmark(ctx, 1);
  ctx->counters.esdf_blocks = ctx->h_state.esdf_blocks;
  if (ctx->h_state.esdf_blocks || robot_pending) {
    if (ctx->h_state.esdf_raise_any || robot_pending) {
      rc = esdf_phase<VPS, FULL>(ctx, e, cr, 0, used, &sweeps, &g_sweep);
      if (rc) return rc;
      if (FULL) KLAUNCH(k_esdf_rotate_active, grid_for(used), dim3(256), 0, s, e, used, 1);
    }
    tmark(ctx, 3);
    rc = esdf_phase<VPS, FULL>(ctx, e, cr, 1, used, &sweeps, &g_sweep);
    if (rc) return rc;
    if (FULL) KLAUNCH(k_esdf_rotate_active, grid_for(used), dim3(256), 0, s, e, used, 1);
    if (!FULL) {  // full-Euclidean parents are part of the state, not a by-product to canonicalise
      rc = esdf_phase<VPS, FULL>(ctx, e, cr, 2, used, &sweeps, &g_sweep);
      if (rc) return rc;
    }
    tmark(ctx, 6);
    if (ctx->h_state.esdf_raise_any || robot_pending) HIP_TRY(hipMemsetAsync(e.raised, 0, nv, s));
  }
  if (clear_updated_flag && !batch)
    KLAUNCH(k_esdf_clear_tsdf_bit, grid_for(used), dim3(256), 0, s, m, e, used);
  tmark(ctx, 7);
  rc = sync_state(ctx);
  if (rc) return rc;
  ctx->counters.esdf_sweeps = sweeps;
  ctx->counters.esdf_relaxations = ctx->h_state.esdf_relax_blocks;
  if (ctx->timing) {
    (void)hipEventSynchronize(ctx->ev[7]);
    vbx_timing& o = ctx->last_timing;
    o = vbx_timing{};
    float t = 0;
    (void)hipEventElapsedTime(&o.total_ms, ctx->ev[0], ctx->ev[7]);
    (void)hipEventElapsedTime(&t, ctx->ev[0], ctx->ev[1]);
    o.prep_ms = t;  // phase 1 (classification)
    if (ctx->ev_hit[3]) { (void)hipEventElapsedTime(&t, ctx->ev[1], ctx->ev[3]); o.solve_ms = t; }  // raise
    if (ctx->ev_hit[6]) { (void)hipEventElapsedTime(&t, ctx->ev[3], ctx->ev[6]); o.fold_ms = t; }   // lower
  }
  return VBX_OK;
}

// EsdfIntegrator::addNewRobotPosition (esdf_integrator.cc:25-92).
int esdf_add_new_robot_position(vbx_ctx* ctx, const vbx_esdf_cfg* cfg, const float position[3]) {
  HIP_TRY(hipSetDevice(ctx->device));
  MapDev& m = ctx->map;
  hipStream_t s = ctx->stream;
  int rc = esdf_ensure(ctx);
  if (rc) return rc;
  EsdfDev e = esdf_dev(ctx);
  HIP_TRY(hipMemsetAsync(&ctx->d_state->new_count, 0, 4, s));
  HIP_TRY(hipMemsetAsync(&ctx->d_state->error, 0, 4, s));
  const float radii[2] = {cfg->clear_sphere_radius, cfg->occupied_sphere_radius};
  const bool ordered = cfg->reference_order != 0;
  if (ordered) {
    if (cfg->num_buckets < 1 || cfg->num_buckets > 254) {
      ctx->fail("addNewRobotPosition (reference order): num_buckets must be in 1..254");
      return VBX_ERR_INVALID;
    }
    if (ctx->esdf_robot_pending && !ctx->esdf_robot_ordered) {
      ctx->fail("addNewRobotPosition: work of an earlier call with reference_order = 0 is pending; update first");
      return VBX_ERR_UNSUPPORTED;
    }
    if (ctx->esdf_robot_pending && (ctx->esdf_robot_buckets != cfg->num_buckets || ctx->esdf_robot_max_distance != cfg->max_distance_m)) {
      ctx->fail("addNewRobotPosition (reference order): num_buckets / max_distance_m changed while entries are queued");
      return VBX_ERR_INVALID;
    }
  } else if (ctx->esdf_robot_pending && ctx->esdf_robot_ordered) {
    ctx->fail("addNewRobotPosition: work of an earlier call with reference_order = 1 is pending; update first");
    return VBX_ERR_UNSUPPORTED;
  }
  for (int pass = 0; pass < 2; ++pass) {
    SphereDev sp;
    // planning_utils_inl.h:18-26: the float loop variable, stepped exactly like the reference's
    sp.r = radii[pass] / m.voxel_size;
    std::vector<float> xs;
    for (float x = -sp.r; x <= sp.r; x++) {
      xs.push_back(x);
      if (xs.size() > 2048) {
        ctx->fail("addNewRobotPosition: sphere radius of more than 1024 voxels");
        return VBX_ERR_INVALID;
      }
    }
    if (xs.empty()) continue;  // negative / NaN radius: empty list
    sp.n = (int)xs.size();
    sp.center = grid_index_from_point(f3{position[0], position[1], position[2]}, m.voxel_size_inv);
    DBuf& bx = pass ? ctx->b_sphere1 : ctx->b_sphere0;
    HIP_TRY(bx.ensure(xs.size() * 4));
    HIP_TRY(hipMemcpyAsync(bx.p, xs.data(), xs.size() * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));  // xs is a stack-lifetime staging buffer
    sp.xs = bx.as<float>();
    const size_t cube = (size_t)sp.n * sp.n * sp.n;
    // Layer::allocateBlockPtrByIndex never fails (layer.h:133-160): a pass that runs out of pool slots doubles
    // the pool and marks its blocks again, like every other allocating call (grow_pool also drops the keys the
    // failed pass left without a slot; a refused growth leaves a consistent map and VBX_ERR_CAPACITY).
    for (;;) {
      KLAUNCH(k_sphere_mark_blocks, grid_for(cube), dim3(256), 0, s, m, sp, ctx->b_newlist.as<uint32_t>(),
                         ctx->d_state);
      KLAUNCH(k_assign_slots, grid_for(m.cap_blocks), dim3(256), 0, s, m, ctx->b_newlist.as<uint32_t>(),
                         ctx->d_state);
      KLAUNCH(k_commit_alloc, dim3(1), dim3(1), 0, s, m, ctx->d_state);
      rc = sync_state(ctx);
      if (rc) return rc;
      if (!(ctx->h_state.error & 1u)) break;
      rc = grow_pool(ctx);
      if (rc) return rc;
      e = esdf_dev(ctx);  // the ESDF arrays moved with the pool
    }
    if (!ordered) {
      KLAUNCH(k_sphere_apply, grid_for(cube), dim3(256), 0, s, m, e, sp, cfg->default_distance_m, pass);
      continue;
    }
    // reference_order: the voxel changes on the device, the containers' order on the host.  block_voxel_list
    // (esdf_integrator.cc:29-34, :60-66) is an unordered_map from block index to the block's voxels in the order of the
    // x / y / z loops of getSphereAroundPoint (planning_utils_inl.h:25-50); a block enters the map when its first voxel
    // comes up.  Walking the cube in that order and inserting into a map with the reference's hash gives its iteration order.
    // (per-call scratch of the sorts: nothing else owns these between calls)
    HIP_TRY(ctx->b_vals1.ensure(cube * 4));
    HIP_TRY(ctx->b_keys1.ensure(cube * 2));
    KLAUNCH(k_sphere_apply_ordered, grid_for(cube), dim3(256), 0, s, m, e, sp, cfg->default_distance_m, cfg->max_distance_m,
            cfg->num_buckets, pass, ctx->b_vals1.as<uint32_t>(), ctx->b_keys1.as<uint16_t>());
    rc = sync_state(ctx);
    if (rc) return rc;
    if (ctx->h_state.error) return check_state_error(ctx);
    const uint32_t used = ctx->h_state.pool_used;
    std::vector<uint32_t> gids(cube);
    std::vector<uint16_t> codes(cube);
    std::vector<int32_t> bidx((size_t)used * 3);
    HIP_TRY(hipMemcpy(gids.data(), ctx->b_vals1.p, cube * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(codes.data(), ctx->b_keys1.p, cube * 2, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(bidx.data(), m.blk_idx, (size_t)used * 12, hipMemcpyDeviceToHost));
    std::unordered_map<HostBlockIdx, std::vector<uint32_t>, HostAnyIndexHash> block_voxel_list;
    uint32_t last_slot = kInvalidSlot;
    std::vector<uint32_t>* last_list = nullptr;
    for (size_t t = 0; t < cube; ++t) {
      if (gids[t] == kInvalidSlot) continue;
      const uint32_t slot = gids[t] / m.nvox;
      if (slot != last_slot) {
        last_slot = slot;
        last_list = &block_voxel_list[HostBlockIdx{bidx[3 * slot], bidx[3 * slot + 1], bidx[3 * slot + 2]}];
      }
      last_list->push_back((uint32_t)t);
    }
    for (const auto& kv : block_voxel_list) {
      for (const uint32_t t : kv.second) {
        const uint32_t code = codes[t];
        if (code & 2u) ctx->esdf_seed_raise.push_back(gids[t]);   // :48 (before the voxel is rewritten; the order is all that matters)
        if (code & 1u) {                                          // :54, :80
          if (ctx->esdf_updated_set.insert(kv.first).second) ctx->esdf_updated_seq.push_back(kv.first);
        } else if (code & 4u) {                                   // :84
          ctx->esdf_seed_open.push_back(gids[t]);
          ctx->esdf_seed_open_bucket.push_back((uint8_t)(code >> 8));
        }
      }
    }
  }
  ctx->esdf_robot_pending = true;
  if (ordered) {
    ctx->esdf_robot_ordered = true;
    ctx->esdf_robot_buckets = cfg->num_buckets;
    ctx->esdf_robot_max_distance = cfg->max_distance_m;
  }
  rc = sync_state(ctx);
  if (rc) return rc;
  return check_state_error(ctx);
}

// ---- the parallel open set of reference_order (vbx_esdf_replay_core.hpp) ------------------------------------------
constexpr int kRpGrid = 1024;       // workgroups of k_rp_step (grid-stride over the phase's items; VBX_RP_GRID: 256 / 512 / 1024 / 2048 / 4096 measured 47.9 / 47.0 / 46.2 / 48.6 / 52.9 ms on frame 11 of the stream)

static uint32_t rp_env_u32(const char* name, uint32_t dflt) {
  const char* v = getenv(name);
  return v ? (uint32_t)strtoul(v, nullptr, 10) : dflt;
}

// lists of ranked excursions the buffers hold: one per base record of a super-step (at least the 4,096 of rounds 4 / 5a)
static uint32_t rp_sub_slots(uint32_t kmax) { return std::max<uint32_t>(kmax, 4096); }

// buffers of the replay, sized for kmax base records per super-step
int rp_ensure(vbx_ctx* ctx, uint32_t num_buckets, size_t n_chunks, uint32_t used, size_t n_list) {
  hipStream_t s = ctx->stream;
  const MapDev& m = ctx->map;
  const uint32_t kmax = rp_env_u32("VBX_RP_KMAX", 16384), smax = std::min<uint32_t>(rp_env_u32("VBX_RP_SMAX", 1024), kSimMax);
  const uint32_t rec_cap = kmax * 12 + 65536, tgt_cap = rec_cap * 2;
  const bool fresh = !ctx->rp_ctl.p;
  HIP_TRY(ctx->rp_ctl.ensure(sizeof(rp::Ctl) + (size_t)rp::kPushShards * (rp::kMaxBuckets + 2) * 4));   // + Args::push_shards behind the block
  if (getenv("VBX_RP_STATS")) {   // what the rankings cost, per workgroup (rp::Args::wg_stats), zero at the start of an update
    HIP_TRY(ctx->rp_wg_stats.ensure((size_t)4096 * (rp::kWgStats + 40) * 8));
    HIP_TRY(hipMemsetAsync(ctx->rp_wg_stats.p, 0, (size_t)4096 * (rp::kWgStats + 40) * 8, s));
  }
  HIP_TRY(ctx->rp_nbslot.ensure((size_t)std::max<uint32_t>(used, 1) * 27 * 4));
  HIP_TRY(ctx->rp_hazard.ensure((size_t)std::max<uint32_t>(used, 1) * m.nvox));
  HIP_TRY(ctx->rp_chunk_tab.ensure((size_t)(num_buckets + 1) * n_chunks * 4));
  if (fresh || ctx->rp_rec_cap != rec_cap || ctx->rp_tgt_cap != tgt_cap || ctx->rp_smax != smax) {
    HIP_TRY(ctx->rp_rec_u32.ensure((size_t)rec_cap * 4 * 12));
    HIP_TRY(ctx->rp_rec_T.ensure((size_t)rec_cap * 8));
    HIP_TRY(ctx->rp_rec_kid.ensure((size_t)rec_cap * 26 * 4));
    HIP_TRY(ctx->rp_rec_tgts.ensure((size_t)rec_cap * 27 * 4));
    HIP_TRY(ctx->rp_rec_push.ensure((size_t)rec_cap * 7 * 4));
    HIP_TRY(ctx->rp_tgt_u32.ensure((size_t)tgt_cap * 4 * 3));
    HIP_TRY(ctx->rp_tgt_ev.ensure((size_t)tgt_cap * rp::kEvMax * 4));   // (sized for the largest Cfg::ev)
    HIP_TRY(ctx->rp_dl.ensure((size_t)tgt_cap * 4 * 2));
    HIP_TRY(ctx->rp_lists.ensure((size_t)rec_cap * 4 * (7 * rp::kShards + 4) + (size_t)kmax * 4));   // changed / born lists (kShards shards each), change points (one per item), dirty excursions
    HIP_TRY(ctx->rp_sub.ensure((size_t)kmax * 4 * 6 + 64 + (size_t)rec_cap * 4));
    const uint32_t sub_slots = rp_sub_slots(kmax);
    HIP_TRY(ctx->rp_sub_list.ensure((size_t)sub_slots * smax * 4));
    HIP_TRY(ctx->rp_sim_q.ensure((size_t)sub_slots * smax * 8));
    HIP_TRY(ctx->rp_ord.ensure((size_t)rec_cap * 4));

    // what the phases expect to be zero between super-steps
    HIP_TRY(hipMemsetAsync(ctx->rp_rec_u32.p, 0, (size_t)rec_cap * 4 * 12, s));
    HIP_TRY(hipMemsetAsync(ctx->rp_rec_kid.p, 0, (size_t)rec_cap * 26 * 4, s));
    HIP_TRY(hipMemsetAsync(ctx->rp_rec_push.p, 0, (size_t)rec_cap * 7 * 4, s));
    HIP_TRY(hipMemsetAsync(ctx->rp_tgt_u32.p, 0, (size_t)tgt_cap * 4 * 3, s));
    HIP_TRY(hipMemsetAsync(ctx->rp_tgt_u32.p, 0xFF, (size_t)tgt_cap * 4, s));   // tgt_gid: kNone — an id nobody took is a hole (PH_CLEANUP keeps it so)
    HIP_TRY(hipMemsetAsync(ctx->rp_sub.p, 0, (size_t)kmax * 4 * 6 + 64 + (size_t)rec_cap * 4, s));
    if (getenv("VBX_RP_POISON")) {   // debug: nothing may depend on what a fresh allocation happens to hold
      DBuf* junk[] = {&ctx->rp_rec_T, &ctx->rp_rec_tgts, &ctx->rp_tgt_ev, &ctx->rp_dl, &ctx->rp_lists, &ctx->rp_sub_list, &ctx->rp_sim_q, &ctx->rp_ord};
      for (DBuf* b : junk) HIP_TRY(hipMemsetAsync(b->p, 0xCD, b->cap, s));
    }
    ctx->rp_rec_cap = rec_cap;
    ctx->rp_tgt_cap = tgt_cap;
    ctx->rp_kmax = kmax;
    ctx->rp_smax = smax;
  }
  {
    // chained-scan descriptors: one per tile of the longest scan (the voxel walk of the classification); ticket = 0, generation = 1
    // (a caller's list may name blocks the map does not hold, esdf_integrator.cc:139-143, or a block twice: it can be
    // longer than the pool, and k_cls_push scans n_list * nvox items)
    const size_t items = std::max<size_t>(rec_cap, std::max<size_t>(std::max<uint32_t>(used, 1), n_list) * m.nvox);
    const size_t bytes = (items / kRpThreads + 2) * rp::kScanC * 8 + 64;
    ctx->rp_scan_tiles_cap = (uint32_t)std::min<size_t>(items / kRpThreads + 2, 0xFFFFFFFFu);
    if (ctx->rp_scan_desc.cap < bytes) {
      HIP_TRY(ctx->rp_scan_desc.ensure(bytes));
      HIP_TRY(hipMemsetAsync(ctx->rp_scan_desc.p, 0, ctx->rp_scan_desc.cap, s));
      const uint32_t t01[2] = {0u, 1u};
      HIP_TRY(hipMemcpyAsync(ctx->rp_scan_desc.p, t01, 8, hipMemcpyHostToDevice, s));
      HIP_TRY(hipStreamSynchronize(s));
    }
  }
  // voxel -> target map over the whole pool (zero between super-steps; a pool that grew gets a zeroed tail)
  const size_t vbytes = (size_t)m.cap_blocks * m.nvox * 4;
  if (ctx->rp_vox2tgt.cap < vbytes) {
    HIP_TRY(ctx->rp_vox2tgt.ensure(vbytes));
    ctx->rp_vox2tgt_zeroed = 0;
  }
  if (ctx->rp_vox2tgt_zeroed < vbytes) {
    HIP_TRY(hipMemsetAsync(ctx->rp_vox2tgt.p, 0, ctx->rp_vox2tgt.cap, s));
    ctx->rp_vox2tgt_zeroed = ctx->rp_vox2tgt.cap;
  }
  return VBX_OK;
}

rp::Args rp_args(vbx_ctx* ctx, const vbx_esdf_cfg* cfg, const EsdfDev& e, size_t n_chunks) {
  const MapDev& m = ctx->map;
  rp::Args a{};
  a.c.max_distance = cfg->max_distance_m;
  a.c.min_diff = cfg->min_diff_m;
  a.c.voxel_size = m.voxel_size;
  a.c.default_distance = cfg->default_distance_m;
  a.c.full = cfg->full_euclidean_distance != 0;
  a.c.multi_queue = cfg->multi_queue != 0;
  a.c.num_buckets = cfg->num_buckets;
  a.c.kmax = ctx->rp_bulk ? std::min<uint32_t>(ctx->rp_kmax, std::max<uint32_t>(64u, rp_env_u32("VBX_RP_KMAX_BULK", 12288))) : ctx->rp_kmax;   // (a bulk update's super-steps: fewer base records at a time)
  a.c.smax = ctx->rp_smax;
  a.c.max_iters = rp_env_u32("VBX_RP_MAX_ITERS", ctx->rp_bulk ? 64 : 128);   // (vbx_ctx::rp_bulk)
  a.ctl = ctx->rp_ctl.as<rp::Ctl>();
  a.push_shards = reinterpret_cast<uint32_t*>(a.ctl + 1);
  a.dist = e.dist;
  a.state = e.state;
  a.nbslot = ctx->rp_nbslot.as<uint32_t>();
  a.hazard = ctx->rp_hazard.as<uint8_t>();
  a.c.filter = rp_env_u32("VBX_RP_FILTER", 3);
  a.c.ev = std::min<uint32_t>(std::max<uint32_t>(rp_env_u32("VBX_RP_EV", 512), 32), rp::kEvMax);   // events per target (128 until round 5a, 256 until the end of round 6: a list that fills up poisons the record that does not fit and cuts the super-step)
  // (after a cut: a bulk update's cuts come in clusters — full event lists in crowded space — and twice what got through is
  // the size that passes; an incremental update's rare cuts are single excursions that outgrew smax, and the bucket goes on
  // at full size: first update 394 / 452 / 517 ms at 2 / 16 / 64, mean of updates 3-22 24.5 / 24.2 / 24.1 ms)
  a.c.cut_mult = std::max<uint32_t>(1u, rp_env_u32("VBX_RP_CUT_MULT", ctx->rp_bulk ? 2 : 64));
  a.c.ramp_mult = std::max<uint32_t>(2u, rp_env_u32("VBX_RP_RAMP_MULT", 8));
  a.c.fold_pairs = rp_env_u32("VBX_RP_FOLD_PAIRS", ctx->rp_bulk ? 0 : 1);   // (a bulk update's lists are long: more than half of its pairs would be folded singly after all — first update 398 -> 405 ms with pairs)
  a.c.lds_counts = rp_env_u32("VBX_RP_LDS_COUNTS", 1);   // (0: one atomic per push and per target on Ctl::push_cnt / st_relax — rounds 4 / 5a)
  a.c.tgt_claim = rp_env_u32("VBX_RP_TGT_CLAIM", 1);     // (0: a target id is taken before the voxel is known to be free: lost races leave holes)
  a.c.fold_all = rp_env_u32("VBX_RP_FOLD_ALL", 1);       // (0: PH_PLACE_BASE fills the dirty list like every other phase)
  a.c.mark_moved = rp_env_u32("VBX_RP_MARK_MOVED", 2);   // (0: rankings do not mark the targets of the records they moved — rounds 4 / 5a)
  a.blk_dirty = m.blk_flags;
  a.dirty_bit = kFlagEsdfDirty;
  a.nvox = m.nvox;
  a.vps = m.vps;
  a.arena = ctx->b_keys0.as<uint32_t>();
  a.chunk_tab = ctx->rp_chunk_tab.as<uint32_t>();
  a.max_chunks = (uint32_t)n_chunks;
  const uint32_t R = ctx->rp_rec_cap, T = ctx->rp_tgt_cap, K = ctx->rp_kmax;
  a.rec_cap = R;
  uint32_t* ru = ctx->rp_rec_u32.as<uint32_t>();
  a.rec_vox = ru; a.rec_pusher = ru + (size_t)R; a.rec_base = ru + (size_t)2 * R; a.rec_meta = ru + (size_t)3 * R;
  a.rec_meta_n = ru + (size_t)4 * R; a.rec_poison = ru + (size_t)5 * R; a.rec_s = ru + (size_t)6 * R; a.rec_s_n = ru + (size_t)7 * R;
  a.rec_d = reinterpret_cast<float*>(ru + (size_t)8 * R); a.rec_d_n = reinterpret_cast<float*>(ru + (size_t)9 * R);
  a.rec_born_it = ru + (size_t)10 * R;
  a.rec_plocal = ru + (size_t)11 * R;
  a.c.tgt_shards = std::min<uint32_t>(std::max<uint32_t>(rp_env_u32("VBX_RP_TGT_SHARDS", rp::kTgtShards), 1u), rp::kTgtShards);
  a.c.stats = getenv("VBX_RP_STATS") ? 1u : 0u;
  a.wg_stats = (a.c.stats && ctx->rp_wg_stats.p) ? ctx->rp_wg_stats.as<unsigned long long>() : nullptr;
  a.rec_T = ctx->rp_rec_T.as<unsigned long long>();
  a.rec_kid = ctx->rp_rec_kid.as<uint32_t>();
  a.rec_tgts = ctx->rp_rec_tgts.as<uint32_t>();
  a.rec_push = ctx->rp_rec_push.as<uint32_t>();
  a.tgt_cap = T;
  a.vox2tgt = ctx->rp_vox2tgt.as<uint32_t>();
  uint32_t* tu = ctx->rp_tgt_u32.as<uint32_t>();
  a.tgt_gid = tu; a.tgt_cnt = tu + (size_t)T; a.tgt_dirty = tu + (size_t)2 * T;
  a.tgt_ev = ctx->rp_tgt_ev.as<uint32_t>();
  a.dl[0] = ctx->rp_dl.as<uint32_t>(); a.dl[1] = a.dl[0] + (size_t)T;
  uint32_t* l = ctx->rp_lists.as<uint32_t>();
  a.chg = l; a.born = l + (size_t)rp::kShards * R; a.cp = l + (size_t)7 * rp::kShards * R; a.sd_list = l + (size_t)(7 * rp::kShards + 4) * R;
  uint32_t* su = ctx->rp_sub.as<uint32_t>();
  a.sub_dirty = su; a.sub_n = su + (size_t)K; a.sub_slot = su + (size_t)2 * K; a.off0 = su + (size_t)3 * K;
  a.sub_slots_used = su + (size_t)4 * K;
  a.sub_mem_n = su + (size_t)4 * K + 16;
  a.sub_restart = su + (size_t)5 * K + 16;
  a.rec_local = su + (size_t)6 * K + 16;
  a.sub_mem = ctx->rp_sub_list.as<uint32_t>();   // (the device ranks from the member lists; sub_list is the serial form's)
  a.sub_list = ctx->rp_sub_list.as<uint32_t>();
  a.sim_q = ctx->rp_sim_q.as<unsigned long long>();
  // VBX_RP_SLOT_BY_BASE (default 1): base record i's excursion is ranked in list i + 1; 0: 4,096 lists handed out as excursions appear
  a.c.slot_by_base = rp_env_u32("VBX_RP_SLOT_BY_BASE", 1);
  a.sub_slots_cap = a.c.slot_by_base ? rp_sub_slots(ctx->rp_kmax) : 4096;
  a.ord = ctx->rp_ord.as<uint32_t>();
  return a;
}

// runs the replay to the end of open_; the queue (arena, chunk table, heads / tails in the control block) is what
// k_esdf_strict(stop_before_open) left
int rp_run(vbx_ctx* ctx, const rp::Args& a, unsigned long long* pops, unsigned long long* relax) {
  hipStream_t s = ctx->stream;
  RpScan sc;
  sc.desc = ctx->rp_scan_desc.as<unsigned long long>() + 8;
  sc.ticket = ctx->rp_scan_desc.as<uint32_t>();
  sc.max_tiles = ctx->rp_scan_tiles_cap;
  const bool serial = rp_env_u32("VBX_RP_SERIAL", 0) != 0;   // debug: the emulated thread-per-item phases on the device
  rp::Args as = a;   // (serial form: no member lists, ranking from the child table)
  as.sub_mem = nullptr; as.sub_restart = nullptr;
  KLAUNCH(k_rp_begin, dim3(1), dim3(1), 0, s, a);
  const uint32_t rp_grid = std::max<uint32_t>(64u, rp_env_u32("VBX_RP_GRID", kRpGrid));   // (measurement switch)
  // The host does not know how many steps an update takes: it queues batches of kRpGraphSteps launches and looks at Ctl::done.
  // A look does not drain the stream (until round 6 every fourth batch ended in a synchronize, and the device sat idle until the
  // host had seen it and queued the next batch): behind every batch the two words are copied into page-locked memory and an
  // event is recorded; the host waits for the look of batch g - 2 before it queues batch g, so two batches are always queued
  // behind the one that is running, and an update ends with at most two batches of launches that find nothing to do.
  HIP_TRY(ctx->rp_h_done.ensure(64));
  volatile uint32_t* h_done = ctx->rp_h_done.as<uint32_t>();   // [look & 1][phase, done]
  for (int k = 0; k < 2; ++k)
    if (!ctx->rp_look_ev[k]) HIP_TRY(hipEventCreateWithFlags(&ctx->rp_look_ev[k], hipEventDisableTiming));
  h_done[1] = h_done[3] = 0;
  const uint64_t max_graphs = 1u << 20;
  const bool sync_debug = getenv("VBX_RP_SYNC") != nullptr;
  // The batch as ONE graph of kRpGraphSteps kernel nodes with their arguments by value (VBX_RP_GRAPH=0: 64 launches),
  // rebuilt when the arguments change (a handful of times in a map's life): one host call per batch instead of 64
  static const bool use_graph = rp_env_u32("VBX_RP_GRAPH", 1) != 0;
  const bool graph_batch = use_graph && !serial && !ctx->prof && !getenv("VBX_RP_SYNC");
  if (graph_batch) {
    std::vector<unsigned char> key(sizeof(rp::Args) + sizeof(RpScan) + 4);
    std::memcpy(key.data(), &a, sizeof(rp::Args));
    std::memcpy(key.data() + sizeof(rp::Args), &sc, sizeof(RpScan));
    std::memcpy(key.data() + sizeof(rp::Args) + sizeof(RpScan), &rp_grid, 4);
    if (!ctx->rp_graph_exec || key != ctx->rp_graph_key) {
      const auto t_g0 = std::chrono::steady_clock::now();
      if (ctx->rp_graph_exec) { (void)hipGraphExecDestroy(ctx->rp_graph_exec); ctx->rp_graph_exec = nullptr; }
      if (ctx->rp_graph) { (void)hipGraphDestroy(ctx->rp_graph); ctx->rp_graph = nullptr; }
      HIP_TRY(hipGraphCreate(&ctx->rp_graph, 0));
      hipGraphNode_t prev = nullptr;
      rp::Args ga = a;
      RpScan gsc = sc;
      for (uint32_t i = 0; i < kRpGraphSteps; ++i) {
        uint32_t seq = i;
        void* params[3] = {&ga, &gsc, &seq};
        hipKernelNodeParams kp{};
        kp.func = reinterpret_cast<void*>(&k_rp_step<false>);
        kp.gridDim = dim3(rp_grid);
        kp.blockDim = dim3(kRpThreads);
        kp.sharedMemBytes = 0;
        kp.kernelParams = params;
        kp.extra = nullptr;
        hipGraphNode_t node = nullptr;
        HIP_TRY(hipGraphAddKernelNode(&node, ctx->rp_graph, prev ? &prev : nullptr, prev ? 1 : 0, &kp));
        prev = node;
      }
      HIP_TRY(hipGraphInstantiate(&ctx->rp_graph_exec, ctx->rp_graph, nullptr, nullptr, 0));
      ctx->rp_graph_key = key;
      if (getenv("VBX_RP_ALLOC_MS")) fprintf(stderr, "[rp] batch graph built in %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_g0).count());
    }
  }
  for (uint64_t g = 0; g < max_graphs; ++g) {
    if (g >= 2) {
      HIP_TRY(hipEventSynchronize(ctx->rp_look_ev[g & 1]));   // the look behind batch g - 2
      if (h_done[(g & 1) * 2 + 1]) break;
    }
    if (graph_batch) {
      HIP_TRY(hipGraphLaunch(ctx->rp_graph_exec, s));
      HIP_TRY(hipMemcpyAsync(const_cast<uint32_t*>(h_done) + (g & 1) * 2, &a.ctl->phase, 8, hipMemcpyDeviceToHost, s));   // phase, done
      HIP_TRY(hipEventRecord(ctx->rp_look_ev[g & 1], s));
      continue;
    }
    for (uint32_t i = 0; i < kRpGraphSteps; ++i) {
      if (sync_debug) {   // debug: which phase faults
        rp::Ctl hc;
        (void)hipMemcpy(&hc, a.ctl, sizeof(rp::Ctl), hipMemcpyDeviceToHost);
        fprintf(stderr, "[rp-sync] step %llu phase %u n %u K %u b %u n_rec %u n_tgt %u chg %u born %u n_sd %u n_cp %u iter %u\n", hc.st_steps, hc.phase, hc.n_threads, hc.K,
                hc.bucket, hc.n_rec, hc.n_tgt, hc.a_chg, hc.a_born, hc.n_sd, hc.n_cp, hc.iter);
      }
      if (serial) KLAUNCH(k_rp_step<true>, dim3(rp_grid), dim3(kRpThreads), 0, s, as, sc, i);
      else KLAUNCH(k_rp_step<false>, dim3(rp_grid), dim3(kRpThreads), 0, s, a, sc, i);
      if (sync_debug && hipStreamSynchronize(s) != hipSuccess) { fprintf(stderr, "[rp-sync] fault\n"); }
    }
    HIP_TRY(hipMemcpyAsync(const_cast<uint32_t*>(h_done) + (g & 1) * 2, &a.ctl->phase, 8, hipMemcpyDeviceToHost, s));   // phase, done
    HIP_TRY(hipEventRecord(ctx->rp_look_ev[g & 1], s));
  }
  rp::Ctl hc;
  HIP_TRY(hipMemcpyAsync(&hc, a.ctl, sizeof(rp::Ctl), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (getenv("VBX_RP_STATS")) {
    static const char* names[] = {"done", "begin", "place", "fold", "apply", "sim", "mincut", "cfold", "rank", "rwrite", "push", "cleanup", "raise"};
    fprintf(stderr, "[rp] raise pops %llu in %llu steps\n", hc.st_raise_pops, hc.st_raise_steps);
    fprintf(stderr, "[rp] pops %llu relax %llu supersteps %llu iters %llu folds %llu exc %llu cuts(iters %llu smax %llu) steps %llu poison %llu error %u\n[rp] steps:",
            hc.st_pops, hc.st_relax, hc.st_supersteps, hc.st_iters, hc.st_folds, hc.st_exc, hc.st_cut_iters, hc.st_cut_smax, hc.st_steps,
            hc.st_poison, hc.error);
    for (int k = 1; k < 13; ++k) fprintf(stderr, " %s %llu(%llu, %.2f ms)", names[k], hc.st_phase_steps[k], hc.st_phase_threads[k], hc.st_phase_ticks[k] * 1e-5);
    fprintf(stderr, "\n[rp] control step, us per step: copy in %.2f, statistics %.2f, rp_control %.2f", hc.st_steps ? hc.st_ctl_ticks[0] * 0.01 / hc.st_steps : 0.0,
            hc.st_steps ? hc.st_ctl_ticks[2] * 0.01 / hc.st_steps : 0.0, hc.st_steps ? hc.st_ctl_ticks[1] * 0.01 / hc.st_steps : 0.0);
    {
      static const char* rows[] = {"fold", "apply", "sim", "place", "push", "cfold", "cleanup", "raise"};
      for (int r = 0; r < 8; ++r) {
        fprintf(stderr, "\n[rp] %s launches by items (<4 <16 <64 <256 <1Ki <4Ki <16Ki more): ", rows[r]);
        for (int b = 0; b < 8; ++b) fprintf(stderr, " %llu x %.1f us", hc.st_bin_steps[r][b], hc.st_bin_steps[r][b] ? hc.st_bin_ticks[r][b] * 0.01 / hc.st_bin_steps[r][b] : 0.0);
      }
    }
    if (a.wg_stats) {
      std::vector<unsigned long long> ws((size_t)4096 * (rp::kWgStats + 40)), tot(rp::kWgStats, 0), ft(8, 0), pt(8, 0);
      HIP_TRY(hipMemcpy(ws.data(), a.wg_stats, ws.size() * 8, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < (size_t)4096 * rp::kWgStats; ++i) tot[i % rp::kWgStats] += ws[i];
      for (size_t i = 0; i < (size_t)4096 * 32; ++i) ft[i % 8] += ws[(size_t)4096 * rp::kWgStats + i];
      for (size_t i = 0; i < (size_t)4096 * 8; ++i) pt[i % 8] += ws[(size_t)4096 * (rp::kWgStats + 32) + i];
      fprintf(stderr, "\n[rp] push tiles: %llu (%.1f buckets per pass); us per tile: ticket %.2f, counts %.2f, scan %.2f, look-back %.2f, apply %.2f; last ticket %.2f us per workgroup-launch",
              pt[6], pt[6] ? (double)pt[7] / pt[6] : 0.0, pt[6] ? pt[0] * 0.01 / pt[6] : 0.0, pt[6] ? pt[1] * 0.01 / pt[6] : 0.0, pt[6] ? pt[2] * 0.01 / pt[6] : 0.0,
              pt[6] ? pt[3] * 0.01 / pt[6] : 0.0, pt[6] ? pt[4] * 0.01 / pt[6] : 0.0, pt[6] ? pt[5] * 0.01 / pt[6] : 0.0);
      fprintf(stderr, "\n[rp] folds: %llu, events %llu, lists of 64 or more %llu, of 192 or more %llu; wave-ms summed over folds: loads %.2f, order %.2f, replay %.2f, outputs %.2f",
              ft[0], ft[1], ft[6], ft[7], ft[2] * 1e-5, ft[3] * 1e-5, ft[4] * 1e-5, ft[5] * 1e-5);
      {
        const unsigned long long slow = tot[19] + tot[20] + tot[21];
        fprintf(stderr, "\n[rp] rankings by duration (<8 <12 <16 <24 <32 us, more): %llu %llu %llu %llu %llu %llu; the %llu of 16 us or more: loads %.2f us, tables and queue %.2f, "
                "replay %.2f, write-back %.2f each, %.0f members, %.1f batches, %.0f pops", tot[16], tot[17], tot[18], tot[19], tot[20], tot[21], slow,
                slow ? tot[22] * 0.01 / slow : 0.0, slow ? tot[23] * 0.01 / slow : 0.0, slow ? tot[24] * 0.01 / slow : 0.0, slow ? tot[25] * 0.01 / slow : 0.0,
                slow ? (double)tot[26] / slow : 0.0, slow ? (double)tot[27] / slow : 0.0, slow ? (double)tot[28] / slow : 0.0);
      }
      fprintf(stderr, "\n[rp] rankings: %llu, members loaded %llu, pops replayed %llu, by pops replayed <16: %llu <64: %llu <256: %llu more: %llu", tot[15], tot[0], tot[1],
              tot[2], tot[3], tot[4], tot[5]);
      fprintf(stderr, "; workgroup-ms summed over rankings: header %.2f, records %.2f, tables %.2f, queue at the restart point %.2f, queue replay %.2f, pop times written %.2f, "
              "marks %.2f, filed %.2f, batches %llu", tot[6] * 1e-5, tot[7] * 1e-5, tot[8] * 1e-5, tot[9] * 1e-5, tot[10] * 1e-5, tot[11] * 1e-5, tot[12] * 1e-5, tot[13] * 1e-5, tot[14]);
    }
    fprintf(stderr, "\n");
  }
  if (!hc.done) {
    ctx->fail("ESDF reference order: the replay did not finish");
    return VBX_ERR_HIP;
  }
  if (hc.error) {
    ctx->fail("ESDF reference order: replay error 0x%x (1 records, 2 targets, 4 queue arena, 8 no progress, 16 event list at the first record, 64 scan wait)", hc.error);
    return (hc.error & ~(8u | 64u)) ? VBX_ERR_CAPACITY : VBX_ERR_HIP;
  }
  *pops = hc.st_pops + hc.st_raise_pops;
  *relax = hc.st_relax;
  return VBX_OK;
}

// updateFromTsdfBlocks' voxel loop in parallel (vbx_kernels_esdf_classify.hpp).  Returns 1 when it has filled the queues,
// 0 when the list names a block twice or the neighbour looks did not settle (the one-wave form takes over; nothing has
// been written to the layer then), < 0 on error.
int esdf_classify_parallel(vbx_ctx* ctx, const EsdfCfgDev& c, const EsdfDev& e, int incremental, int batch_crust, int num_buckets,
                           const uint32_t* list_slots, uint32_t n_list, uint32_t used, const rp::Args& ra, unsigned long long* n_blocks) {
  hipStream_t s = ctx->stream;
  const MapDev& m = ctx->map;
  *n_blocks = 0;
  const size_t nv = (size_t)std::max<uint32_t>(n_list, 1) * m.nvox;
  HIP_TRY(ctx->cls_pos.ensure((size_t)std::max<uint32_t>(used, 1) * 4));
  HIP_TRY(ctx->cls_nb27.ensure((size_t)std::max<uint32_t>(n_list, 1) * 27 * 4));
  HIP_TRY(ctx->cls_shadow.ensure(nv * 10));
  HIP_TRY(ctx->cls_counters.ensure(4 * kClsCounters));
  HIP_TRY(hipMemsetAsync(ctx->cls_pos.p, 0xFF, (size_t)std::max<uint32_t>(used, 1) * 4, s));
  HIP_TRY(hipMemsetAsync(ctx->cls_counters.p, 0, 4 * kClsCounters, s));
  ClsArgs a{};
  a.m = m; a.e = e; a.c = c;
  a.incremental = incremental; a.batch_crust = batch_crust; a.num_buckets = num_buckets;
  a.list_slots = list_slots; a.n_list = n_list;
  a.slot_pos = ctx->cls_pos.as<uint32_t>();
  a.nb27 = ctx->cls_nb27.as<uint32_t>();
  a.sh_d = ctx->cls_shadow.as<float>();
  a.sh_s = reinterpret_cast<uint32_t*>(a.sh_d + nv);
  a.sh_f = reinterpret_cast<uint8_t*>(a.sh_s + nv);
  a.sh_q = a.sh_f + nv;
  a.counters = ctx->cls_counters.as<uint32_t>();
  uint32_t h[4] = {0, 0, 0, 0};
  if (n_list) {
    KLAUNCH(k_cls_positions, grid_for(n_list), dim3(256), 0, s, a);
    KLAUNCH(k_cls_neighbours, grid_for((size_t)n_list * 27), dim3(256), 0, s, a);
    KLAUNCH(k_cls_base, grid_for((size_t)n_list * m.nvox), dim3(256), 0, s, a);
    int rounds = 0;
    for (;; ++rounds) {
      if (incremental) {
        switch (m.vps) {
          case 8: KLAUNCH(k_cls_nb_block<8>, dim3(n_list), dim3(64), 0, s, a); break;
          case 16: KLAUNCH(k_cls_nb_block<16>, dim3(n_list), dim3(256), 0, s, a); break;
          default: return 0;   // (other block sizes: the one-wave walk)
        }
      }
      HIP_TRY(hipMemcpyAsync(h, a.counters, 16, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      if (getenv("VBX_RP_STATS")) fprintf(stderr, "[cls] round %d: moved %u (dup %u, blocks %u)\n", rounds, h[1], h[0], h[2]);
      if (h[0] != 0) return 0;             // a block listed twice
      if (!incremental || h[1] == 0) break;
      if (rounds >= 256) return 0;         // (values crossing block faces for longer than any map's walk)
      HIP_TRY(hipMemsetAsync(a.counters + 1, 0, 4, s));
    }
    KLAUNCH(k_cls_commit, grid_for((size_t)n_list * m.nvox), dim3(256), 0, s, a);
  }
  KLAUNCH(k_cls_reserve, dim3(1), dim3(1), 0, s, a, ra);
  if (n_list) {
    RpScan sc;
    sc.desc = ctx->rp_scan_desc.as<unsigned long long>() + 8;
    sc.ticket = ctx->rp_scan_desc.as<uint32_t>();
    sc.max_tiles = ctx->rp_scan_tiles_cap;
    const uint32_t tiles = (uint32_t)(((size_t)n_list * m.nvox + kRpThreads - 1) / kRpThreads);
    for (int q0 = 0; q0 <= num_buckets; q0 += rp::kScanC)
      KLAUNCH(k_cls_push, dim3(std::min<uint32_t>(tiles, 1024u)), dim3(kRpThreads), 0, s, a, ra, sc, q0);
  }
  *n_blocks = h[2];
  return 1;
}

// cfg->reference_order: the whole update as ONE sequential replay on the device (vbx_kernels_esdf_strict.hpp).
int esdf_update_strict(vbx_ctx* ctx, const vbx_esdf_cfg* cfg, int batch, int clear_updated_flag, const int32_t* list,
                       size_t n_list, int list_incremental) {
  MapDev& m = ctx->map;
  hipStream_t s = ctx->stream;
  if (cfg->num_buckets < 1 || cfg->num_buckets > kStrictMaxBuckets) {
    ctx->fail("ESDF reference order: num_buckets must be in 1..%d", kStrictMaxBuckets);  // CHECK_NE(num_buckets_, 0), bucket_queue.h:42
    return VBX_ERR_INVALID;
  }
  int rc = esdf_ensure(ctx);
  if (rc) return rc;
  rc = sync_state(ctx);
  if (rc) return rc;
  const uint32_t used = ctx->h_state.pool_used;
  ctx->counters = vbx_counters{};
  if (used == 0) return VBX_OK;
  if (ctx->esdf_robot_pending && !ctx->esdf_robot_ordered && !batch) {
    ctx->fail("ESDF reference order: addNewRobotPosition left its work as order-free marks (that call had reference_order = 0); "
              "run this update with reference_order = 0 too");
    return VBX_ERR_UNSUPPORTED;
  }
  // what addNewRobotPosition left in raise_ / open_ (esdf_integrator.cc:48, :84) is in front of everything this update
  // pushes.  A batch update drops the ESDF layer the entries point into (the reference would die on them,
  // esdf_integrator.cc:379-381 CHECK_NOTNULL): they are dropped with it.
  const int NB = cfg->num_buckets;
  std::vector<std::vector<uint32_t>> seeds;
  size_t n_seed = 0;
  if (ctx->esdf_robot_pending && ctx->esdf_robot_ordered && !batch && (!ctx->esdf_seed_raise.empty() || !ctx->esdf_seed_open.empty())) {
    if (ctx->esdf_robot_buckets != NB || ctx->esdf_robot_max_distance != cfg->max_distance_m) {
      ctx->fail("ESDF reference order: num_buckets / max_distance_m differ from the ones addNewRobotPosition queued its entries with");
      return VBX_ERR_INVALID;
    }
    // (an ESDF block removed since — vbx_blocks_remove, vbx_remove_distant_blocks — takes its entries along)
    std::vector<uint32_t> flags(used);
    HIP_TRY(hipMemcpy(flags.data(), m.blk_flags, (size_t)used * 4, hipMemcpyDeviceToHost));
    auto alive = [&](uint32_t gid) { const uint32_t sl = gid / m.nvox; return sl < used && !(flags[sl] & kFlagFree) && (flags[sl] & kFlagEsdfAlloc); };
    seeds.resize((size_t)NB + 1);
    for (size_t i = 0; i < ctx->esdf_seed_open.size(); ++i)
      if (alive(ctx->esdf_seed_open[i])) seeds[std::min<int>(ctx->esdf_seed_open_bucket[i], NB - 1)].push_back(ctx->esdf_seed_open[i]);
    for (const uint32_t g : ctx->esdf_seed_raise)
      if (alive(g)) seeds[NB].push_back(g);
    for (const auto& v : seeds) n_seed += v.size();
  }
  // updated_blocks_ goes behind the TSDF blocks of the update (:98-100, :107-109), in the set's iteration order;
  // updateFromTsdfBlocks(list) leaves the set alone (the caller composes the list: vbx_esdf_robot_updated_blocks)
  std::vector<HostBlockIdx> robot_blocks;
  if (!list) {
    robot_blocks.assign(ctx->esdf_updated_set.begin(), ctx->esdf_updated_set.end());
    ctx->esdf_updated_set.clear();
    ctx->esdf_updated_seq.clear();
  }
  ctx->esdf_robot_pending = false;
  ctx->esdf_robot_ordered = false;
  ctx->esdf_seed_raise.clear();
  ctx->esdf_seed_open.clear();
  ctx->esdf_seed_open_bucket.clear();
  EsdfDev e = esdf_dev(ctx);
  const size_t nv = (size_t)used * m.nvox;
  for (int i = 0; i < 9; ++i) ctx->ev_hit[i] = false;
  tmark(ctx, 0);
  if (batch) {  // esdf_layer_->removeAllBlocks(), esdf_integrator.cc:95
    HIP_TRY(hipMemsetAsync(e.dist, 0, nv * 4, s));
    HIP_TRY(hipMemsetAsync(e.state, 0, nv * 4, s));
    HIP_TRY(hipMemsetAsync(e.raised, 0, nv, s));
  }
  KLAUNCH(k_esdf_reset_flags, grid_for(used), dim3(256), 0, s, m, e, used, batch ? 1 : 0, list ? 1 : 0, ctx->d_state);
  // the blocks in visiting order -> pool slots
  std::vector<uint32_t> h_slots;
  size_t n = 0;
  HIP_TRY(ctx->b_rank.ensure(std::max<size_t>(std::max<size_t>(n_list, used), 1) * 4));
  if (list) {
    n = n_list;
    HIP_TRY(ctx->b_head.ensure(std::max<size_t>(n, 1) * 12));
    HIP_TRY(hipMemcpyAsync(ctx->b_head.p, list, n * 12, hipMemcpyHostToDevice, s));
    if (n)
      KLAUNCH(k_lookup_slots, grid_for(n), dim3(256), 0, s, m, ctx->b_head.as<int32_t>(), (uint32_t)n, 1,
                         ctx->b_rank.as<uint32_t>());
    HIP_TRY(hipStreamSynchronize(s));  // `list` is a caller-owned host buffer
  } else {
    // getAllAllocatedBlocks (batch, :97) / getAllUpdatedBlocks(Update::kEsdf) (:106): ascending (z,y,x) here
    std::vector<uint32_t> flags(used);
    std::vector<int32_t> idx((size_t)used * 3);
    HIP_TRY(hipMemcpy(flags.data(), m.blk_flags, (size_t)used * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(idx.data(), m.blk_idx, (size_t)used * 12, hipMemcpyDeviceToHost));
    std::vector<std::pair<uint64_t, uint32_t>> v;
    for (uint32_t sl = 0; sl < used; ++sl) {
      if (!(flags[sl] & kFlagPublished)) continue;
      if (!batch && !(flags[sl] & 4u)) continue;
      v.emplace_back(pack_block_key(idx[3 * sl], idx[3 * sl + 1], idx[3 * sl + 2]), sl);
    }
    std::sort(v.begin(), v.end());
    // the iteration order of the reference's block_map_, as far as the library saw the Layer being built (integrate calls,
    // uploads); blocks of unknown provenance follow in ascending (z,y,x)
    bool order_exact = true;
    size_t n_unknown = 0;
    rc = order_like_layer(ctx, &v, &order_exact, &n_unknown);
    if (rc) return rc;
    if (!order_exact && !v.empty()) {
      // not the reference Layer's order: say so instead of silently walking another sequence (vbx_counters.esdf_order_inexact;
      // one line on stderr per handle)
      ctx->counters.esdf_order_inexact = std::max<size_t>(n_unknown, 1);
      if (!ctx->warned_esdf_order) {
        ctx->warned_esdf_order = true;
        fprintf(stderr, "[vbx] vbx_esdf_update(reference_order = 1): the place of %zu of %zu blocks in the reference Layer's iteration "
                        "order is unknown to the library (merged in / deserialised / integrated with block-order tracking off / log "
                        "overflow); they are walked in ascending (z,y,x) order behind the others, the result may differ from a "
                        "single-threaded reference run (reported once; vbx_counters.esdf_order_inexact; pass the order with "
                        "vbx_esdf_update_blocks)\n", n_unknown, v.size());
      }
    }
    h_slots.resize(v.size());
    for (size_t i = 0; i < v.size(); ++i) h_slots[i] = v[i].second;
    if (!robot_blocks.empty()) {
      std::unordered_map<uint64_t, uint32_t> slot_of;
      for (uint32_t sl = 0; sl < used; ++sl)
        if (!(flags[sl] & kFlagFree)) slot_of[pack_block_key(idx[3 * sl], idx[3 * sl + 1], idx[3 * sl + 2])] = sl;
      for (const HostBlockIdx& b : robot_blocks) {
        const auto it = slot_of.find(pack_block_key(b.x, b.y, b.z));
        h_slots.push_back(it == slot_of.end() ? kInvalidSlot : it->second);   // (:139-143 skips a block the TSDF layer does not have)
      }
      HIP_TRY(ctx->b_rank.ensure(h_slots.size() * 4));
    }
    n = h_slots.size();
    if (n) HIP_TRY(hipMemcpyAsync(ctx->b_rank.p, h_slots.data(), n * 4, hipMemcpyHostToDevice, s));
  }
  // queue arena: a voxel sits in open_ at most once at a time without multi_queue and in raise_ at most once per
  // update, so 2 x voxels bounds what is queued at once; multi_queue can hold more (loud failure if it does)
  // (chunks are not reused inside an update: the arena holds every push of the update)
  const size_t n_chunks = std::max<size_t>(1024, (size_t)(cfg->multi_queue ? 16 : 4) * nv / kSqChunk + (size_t)4 * cfg->num_buckets + 64) +
                          n_seed / kSqChunk + (size_t)NB + 2;
  HIP_TRY(ctx->b_keys0.ensure(n_chunks * kSqChunk * 4));
  HIP_TRY(ctx->b_vals0.ensure(64));
  ctx->rp_bulk = n > 2 * ctx->rp_walked_total;   // (a batch rebuild of a map that has been updated before is not one: 10.8 -> 8.9 ms without the bulk settings)
  ctx->rp_walked_total += n;
  const bool replay = rp_env_u32("VBX_ESDF_REPLAY", 1) != 0 && cfg->num_buckets <= 254;   // (a push table entry is one byte: bucket + 1, raise_ = num_buckets)
  const auto t_ens0 = std::chrono::steady_clock::now();
  rc = rp_ensure(ctx, (uint32_t)cfg->num_buckets, n_chunks, used, n);
  if (getenv("VBX_RP_ALLOC_MS")) fprintf(stderr, "[rp] rp_ensure %.2f ms (host)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_ens0).count());
  if (rc) return rc;
  HIP_TRY(hipMemsetAsync(ctx->rp_ctl.p, 0, sizeof(rp::Ctl) + (size_t)rp::kPushShards * (rp::kMaxBuckets + 2) * 4, s));
  if (n_seed) {
    // the queues as addNewRobotPosition left them: chunk table rows, arena image, FIFO tails (both forms of the voxel
    // walk start from the control block's tails / chunk_top instead of empty queues)
    std::vector<uint32_t> image, tails((size_t)NB + 1, 0), reserved((size_t)NB + 1, 0), row;
    uint32_t top = 0;
    rp::Ctl* dctl = ctx->rp_ctl.as<rp::Ctl>();
    for (int q = 0; q <= NB; ++q) {
      const std::vector<uint32_t>& v = seeds[q];
      if (v.empty()) continue;
      const uint32_t chunks = (uint32_t)((v.size() + kSqChunk - 1) / kSqChunk);
      row.resize(chunks);
      for (uint32_t j = 0; j < chunks; ++j) row[j] = top + j;
      image.resize((size_t)(top + chunks) * kSqChunk, 0u);
      std::memcpy(image.data() + (size_t)top * kSqChunk, v.data(), v.size() * 4);
      HIP_TRY(hipMemcpy(ctx->rp_chunk_tab.as<uint32_t>() + (size_t)q * n_chunks, row.data(), (size_t)chunks * 4, hipMemcpyHostToDevice));
      top += chunks;
      tails[q] = (uint32_t)v.size();
      reserved[q] = chunks;
    }
    HIP_TRY(hipStreamSynchronize(s));   // (the memset of the control block above)
    HIP_TRY(hipMemcpy(ctx->b_keys0.p, image.data(), image.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(&dctl->tail[0], tails.data(), tails.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(&dctl->reserved[0], reserved.data(), reserved.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(&dctl->chunk_top, &top, 4, hipMemcpyHostToDevice));
  }
  StrictArgs a{};
  a.m = m;
  a.e = e;
  a.c.max_distance = cfg->max_distance_m;
  a.c.min_distance = cfg->min_distance_m;
  a.c.default_distance = cfg->default_distance_m;
  a.c.min_diff = cfg->min_diff_m;
  a.c.min_weight = cfg->min_weight;
  a.c.add_occupied_crust = cfg->add_occupied_crust != 0;
  a.c.voxel_size = m.voxel_size;
  a.full = cfg->full_euclidean_distance != 0;
  a.multi_queue = cfg->multi_queue != 0;
  a.num_buckets = cfg->num_buckets;
  a.incremental = list ? (list_incremental != 0) : (batch ? 0 : 1);
  a.batch_crust = (!a.incremental && cfg->add_occupied_crust) ? 1 : 0;
  a.list_slots = ctx->b_rank.as<uint32_t>();
  a.n_list = (uint32_t)n;
  a.arena = ctx->b_keys0.as<uint32_t>();
  a.chunk_tab = ctx->rp_chunk_tab.as<uint32_t>();
  a.n_chunks = (uint32_t)std::min<size_t>(n_chunks, 0xFFFFFFF0u);
  a.rctl = ctx->rp_ctl.as<rp::Ctl>();
  a.stop_before_open = replay ? 1 : 0;
  a.stats = ctx->b_vals0.as<unsigned long long>();
  a.max_pops = 1ull << 40;
  unsigned long long rp_pops = 0, rp_relax = 0, cls_blocks = 0;
  int front_done = 0;
  if (replay && rp_env_u32("VBX_RP_CLASSIFY", 1)) {
    const rp::Args ra0 = rp_args(ctx, cfg, e, a.n_chunks);
    // A list may name a block twice — esdf_server calls addNewRobotPosition before every update (esdf_server.cc:219-230) and
    // a sphere block that is also an updated TSDF block is listed by both (:105-109) — and the reference then walks the
    // block a second time over what the first visit left.  The parallel walk handles a list WITHOUT repeats, so the list is
    // cut into maximal repeat-free segments that run one after the other: a segment commits to the layer and appends its
    // pushes behind the queues' tails before the next one starts, which is what the sequential loop does.  (Rounds 3-4 sent
    // any list with a repeat through the one-wave walk: every update of such a server.)
    std::vector<uint32_t> seg_slots;
    if (list) {
      seg_slots.resize(n);
      if (n) HIP_TRY(hipMemcpy(seg_slots.data(), ctx->b_rank.p, n * 4, hipMemcpyDeviceToHost));   // (the stream was drained after the lookup)
    } else {
      seg_slots = h_slots;
    }
    std::unordered_set<uint32_t> in_seg;
    const uint32_t* all_slots = a.list_slots;
    size_t pos = 0;
    front_done = 1;
    // every segment is a pass of its own (a memset over the pool, five launches or more, a stream sync per neighbour round):
    // a list that alternates (A, B, A, B, ...) would take O(n) such passes where the one-wave walk takes the list in one —
    // beyond a handful of segments the whole list goes through that form
    static const uint32_t max_segments = rp_env_u32("VBX_RP_MAX_SEGMENTS", 8);
    {
      uint32_t n_seg = 0;
      for (size_t p2 = 0; p2 < n && n_seg <= max_segments; ++n_seg) {
        in_seg.clear();
        for (; p2 < n; ++p2) {
          const uint32_t sl = seg_slots[p2];
          if (sl != kInvalidSlot && !in_seg.insert(sl).second) break;
        }
      }
      if (n_seg > max_segments) front_done = 0;
    }
    if (front_done) do {
      in_seg.clear();
      size_t end = pos;
      for (; end < n; ++end) {
        const uint32_t sl = seg_slots[end];
        if (sl != kInvalidSlot && !in_seg.insert(sl).second) break;   // (absent blocks may repeat: the walk skips them, :139-143)
      }
      unsigned long long seg_blocks = 0;
      const int r = esdf_classify_parallel(ctx, a.c, e, a.incremental, a.batch_crust, a.num_buckets, all_slots + pos, (uint32_t)(end - pos), used, ra0, &seg_blocks);
      if (r < 0) return r;
      if (r == 0) {   // not this segment's voxel walk in parallel (block size, unsettled looks): the one-wave form walks the rest of the list
        a.list_slots = all_slots + pos;
        a.n_list = (uint32_t)(n - pos);
        front_done = 0;
        break;
      }
      cls_blocks += seg_blocks;
      pos = end;
    } while (pos < n);
  }
  if (!front_done) KLAUNCH(k_esdf_strict, dim3(1), dim3(64), 0, s, a);
  if (replay) {
    // processOpenSet: the parallel replay over the queue the kernel above filled
    KLAUNCH(k_rp_nbslot, grid_for((size_t)used * 27), dim3(256), 0, s, m, used, ctx->rp_nbslot.as<uint32_t>());
    const rp::Args ra = rp_args(ctx, cfg, e, a.n_chunks);
    KLAUNCH(k_rp_hazard, grid_for((size_t)used * m.nvox), dim3(256), 0, s, ra, m.blk_flags, used, ctx->rp_hazard.as<uint8_t>());
    rc = rp_run(ctx, ra, &rp_pops, &rp_relax);
    if (rc) return rc;
  }
  KLAUNCH(k_esdf_mark_unsettled, grid_for(used), dim3(256), 0, s, m, used);
  if (clear_updated_flag && !batch && !list && n)
    KLAUNCH(k_esdf_strict_clear_tsdf_bit, grid_for(n), dim3(256), 0, s, m, ctx->b_rank.as<uint32_t>(), (uint32_t)n);
  tmark(ctx, 7);
  unsigned long long st[8] = {0};
  if (!front_done) {
    HIP_TRY(hipMemcpyAsync(st, a.stats, sizeof(st), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    st[6] += cls_blocks;   // (segments the parallel walk handled before the one-wave form took over)
  } else {
    st[6] = cls_blocks;
  }
  if (st[7] == 1) {
    ctx->fail("ESDF reference order: queue arena exhausted (%zu chunks)", n_chunks);
    return VBX_ERR_CAPACITY;
  }
  if (st[7] != 0) {
    ctx->fail("ESDF reference order: the replay stopped with work left in its queues");
    return VBX_ERR_HIP;
  }
  ctx->counters.esdf_blocks = st[6];
  ctx->counters.esdf_relaxations = st[5] + rp_relax;
  ctx->counters.esdf_sweeps = st[4] + st[3] + rp_pops;  // queue pops (open + raise) take the place of sweeps
  if (ctx->timing) {
    (void)hipEventSynchronize(ctx->ev[7]);
    vbx_timing& o = ctx->last_timing;
    o = vbx_timing{};
    (void)hipEventElapsedTime(&o.total_ms, ctx->ev[0], ctx->ev[7]);
  }
  return VBX_OK;
}

// vbx_esdf_reserve: the ESDF layer's device arrays and, in reference order, the replay's pools (sized by the super-step, not by
// the map: ~1.5 GB) before the first update needs them — 30 ms of hipMalloc in a fresh process, up to 100 ms behind the frees
// of another handle, that the first update of a map otherwise spends with the device idle
int esdf_reserve(vbx_ctx* ctx, const vbx_esdf_cfg* cfg) {
  HIP_TRY(hipSetDevice(ctx->device));
  int rc = esdf_ensure(ctx);
  if (rc || !cfg->reference_order) return rc;
  if (cfg->num_buckets < 1 || cfg->num_buckets > kStrictMaxBuckets) {
    ctx->fail("ESDF reference order: num_buckets must be in 1..%d", kStrictMaxBuckets);
    return VBX_ERR_INVALID;
  }
  rc = sync_state(ctx);
  if (rc) return rc;
  rc = rp_ensure(ctx, (uint32_t)cfg->num_buckets, 1024, ctx->h_state.pool_used, 0);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->stream));   // (the pools' clears)
  return VBX_OK;
}

int esdf_update(vbx_ctx* ctx, const vbx_esdf_cfg* cfg, int batch, int clear_updated_flag,
                const int32_t* list = nullptr, size_t n_list = 0, int list_incremental = 0) {
  HIP_TRY(hipSetDevice(ctx->device));
  if (cfg->reference_order) {
    if (ctx->map.vps < 4 || ctx->map.vps > 32) {
      ctx->fail("ESDF: voxels_per_side out of range");
      return VBX_ERR_UNSUPPORTED;
    }
    if (cfg->full_euclidean_distance && !(cfg->max_distance_m / ctx->map.voxel_size < 120.0f)) {
      ctx->fail("ESDF: full_euclidean_distance needs max_distance_m / voxel_size < 120 (int8 parent vectors)");
      return VBX_ERR_UNSUPPORTED;
    }
    return esdf_update_strict(ctx, cfg, batch, clear_updated_flag, list, n_list, list_incremental);
  }
  if (ctx->esdf_robot_pending && ctx->esdf_robot_ordered) {
    if (!batch) {
      ctx->fail("ESDF: addNewRobotPosition queued its work in the reference's order (that call had reference_order = 1); "
                "run this update with reference_order = 1 too");
      return VBX_ERR_UNSUPPORTED;
    }
    ctx->esdf_robot_forget();   // the batch update drops the layer the entries point into
  }
  const bool full = cfg->full_euclidean_distance != 0;
  if (full && !(cfg->max_distance_m / ctx->map.voxel_size < 120.0f)) {
    // parent vectors are kept as int8 per component (the range Block::serializeToIntegers keeps,
    // block.cc:27-32); a wavefront stops at max_distance_m, so that bounds the components
    ctx->fail("ESDF: full_euclidean_distance needs max_distance_m / voxel_size < 120 (int8 parent vectors)");
    return VBX_ERR_UNSUPPORTED;
  }
  switch (ctx->map.vps) {
    case 8:
      return full ? esdf_update_t<8, true>(ctx, cfg, batch, clear_updated_flag, list, n_list, list_incremental)
                  : esdf_update_t<8, false>(ctx, cfg, batch, clear_updated_flag, list, n_list, list_incremental);
    case 16:
      return full ? esdf_update_t<16, true>(ctx, cfg, batch, clear_updated_flag, list, n_list, list_incremental)
                  : esdf_update_t<16, false>(ctx, cfg, batch, clear_updated_flag, list, n_list, list_incremental);
    default:
      ctx->fail("ESDF: voxels_per_side must be 8 or 16 (LDS tile)");
      return VBX_ERR_UNSUPPORTED;
  }
}

}  // namespace

