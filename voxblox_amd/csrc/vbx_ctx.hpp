// vbx_ctx.hpp — the per-map context behind the opaque vbx_ctx handle
// Part of libvbx_hip.so's single translation unit (included by vbx_hip.hip, in order).

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
// BlockIndex keyed like the reference's AnyIndexHash (block_hash.h:20-32): containers of these iterate in the order the
// reference's HierarchicalIndexMap / IndexSet do, given the same insertions and the same libstdc++
struct HostBlockIdx {
  int32_t x, y, z;
  bool operator==(const HostBlockIdx& o) const { return x == o.x && y == o.y && z == o.z; }
};
struct HostAnyIndexHash {
  size_t operator()(const HostBlockIdx& i) const {
    const size_t sl = 17191, sl2 = sl * sl;
    return static_cast<unsigned int>(static_cast<size_t>(i.x) + static_cast<size_t>(i.y) * sl + static_cast<size_t>(i.z) * sl2);
  }
};

// `static int64_t reset_counter` of FastTsdfIntegrator::integratePointCloud (tsdf_integrator.cc:564): ONE counter for every
// FastTsdfIntegrator of the process, whichever Layer it integrates into — two integrators with clear_checks_every_n_frames > 1
// advance each other's.  Calls on one handle come from one thread (the reference's integratePointCloud is documented not
// thread safe); the atomic only keeps concurrent handles (the sharding's delta maps, one host thread each) from tearing it.
static std::atomic<int64_t> g_fast_reset_counter{0};

struct vbx_ctx {
  int device = 0;
  vbx_map_cfg mcfg{};
  MapDev map{};
  uint32_t hcap = 0;
  uint32_t pool_limit = 0;   // vbx_set_pool_limit: the pool never grows beyond this many blocks (0: no limit)
  uint32_t pool_grown = 0;   // times the pool doubled
  bool warned_time_budget = false;  // Fast: max_integration_time_s overrun reported once
  bool warned_esdf_order = false;   // ESDF: a reference-order walk over blocks of unknown Layer order reported once
  double fast_us_per_point = 0.0;   // Fast with a finite max_integration_time_s: wall time per taken point of the earlier calls
  uint32_t fast_take_limit = ~0u;   // ... points of the taking order the current call takes
  uint32_t fast_prev_taken = 0;     // ... points the previous budgeted call took (the limit moves by at most 2x per frame)
  uint64_t fast_budget_calls = 0;   // budgeted calls so far (the first one is left out of the estimate)
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  DevState* d_state = nullptr;
  DevState h_state{};
  StateMirror* h_mirror = nullptr;  // page-locked, mapped (may stay null: plain copies are used then)
  StateMirror* d_mirror = nullptr;
  uint32_t sync_seq = 0;
  std::string err;

  // The order in which the reference's single-threaded integrators hand new blocks to the Layer (tsdf_integrator.cc:91-147):
  // first-touch ranks per block on the device (MapDev::blk_first), a device log of the blocks every call published, and
  // host containers keyed like the reference's that turn the log into iteration orders.
  DBuf b_blkfirst, b_newlog;
  unsigned long long call_seq = 1, last_call_seq = 0;
  unsigned long long last_new_seq = 0;   // the call whose new blocks `last_new` holds
  uint32_t newlog_pending = 0;        // log entries not read back yet
  uint32_t published_since_clear = 0; // blocks the integrate calls published since the map was last cleared (vbx_clear_keep_slots)
  bool new_flags_live = false;        // kFlagNewThisCall may be set on some block
  bool track_block_order = true;      // vbx_set_block_order_tracking (off: MapDev::blk_first is null, nothing is ranked or logged)
  bool layer_order_exact = true;      // every block of the map went through layer_order in the reference's sequence
  std::unordered_map<HostBlockIdx, int, HostAnyIndexHash> temp_block_map;       // TsdfIntegratorBase::temp_block_map_
  std::unordered_map<HostBlockIdx, uint32_t, HostAnyIndexHash> layer_order;     // Layer<TsdfVoxel>::block_map_ (keys only)
  std::vector<HostBlockIdx> last_new;   // the last integrate call's new blocks in Layer::insertBlock sequence

  // pool / map storage
  DBuf b_hkeys, b_hvals, b_dist, b_weight, b_rgba, b_blkidx, b_blkflags, b_freelist, b_newlist;
  // per-call scratch
  DBuf b_pts, b_cols;                                   // staged host input
  DBuf t_px, t_py, t_pz, t_rgba, t_w, t_flags, t_bkey;  // ray table A (per point, order s)
  DBuf u_px, u_py, u_pz, u_rgba, u_w, u_flags, u_bkey;  // ray table B (bundles / kept rays)
  DBuf w_px, w_py, w_pz, w_rgba, w_w, w_flags, w_bkey;  // ray table C (Merged: bundles in key order, before the order permutation)
  hipEvent_t ev_copy = nullptr;  // Merged: the insertion-order list has reached the host
  DBuf b_pcx, b_pcy, b_pcz;                             // Merged: point_C per s
  DBuf b_cnt, b_off, b_keys0, b_keys1, b_vals0, b_vals1, b_tmp, b_head, b_rank, b_graze;
  DBuf b_fin;  // per-update fold inputs (sdf, weight, colour) in sorted key order
  DBuf b_bstart, b_mgather;  // Merged: first sorted position of every bundle; point data in sorted order
  DBuf b_T, b_TH, b_U, b_vox, b_vhash, b_cl, b_act0, b_act1, b_long, b_order, b_obs, b_sphere0, b_sphere1, b_redo, b_fix, b_bkeys, b_bfirst, b_bperm, b_obsset, b_collided, b_hist0, b_hist1, b_moved;
  DBuf b_scan_desc;  // k_scan_excl: ticket counter + tile descriptors
  DBuf b_fs_hist, b_fs_desc;  // fused radix pass (vbx_sort.hpp): histogram ring; ticket + flags + count rows
  uint32_t fs_ring_pos = 0, fs_ticket_base = 0, fs_gen = 0, fs_desc_tiles = 0;
  int fs_enabled = -1;  // VBX_SORT_FUSED, read once
  DBuf b_bbox;       // Merged: per-workgroup bounding boxes of the endpoint voxels
  uint32_t scan_ticket_base = 0, scan_gen = 0;
  uint32_t obs_epoch = 1;
  uint32_t fast_last_iters = 0;  // sweeps the previous Fast frame needed
  uint32_t fast_redo_grid = 0;   // rays the second list-building pass is launched for
  // Fast integrator persistent state
  DBuf b_startset;       // ApproxHashSet<20,10000> storage (u32 per slot)
  uint32_t start_offset = 0;
  bool start_sentinel_live = true;
  bool startset_init = false;
  std::vector<int32_t> h_by_s;  // scratch of merged_reference_order
  PBuf h_mkeys, h_mperm;                                 // its read-back / upload staging
  PBuf rp_h_done;                                        // ESDF replay: page-locked landing place of the looks at Ctl::done (two in flight)
  hipEvent_t rp_look_ev[2] = {nullptr, nullptr};
  // a batch of the replay's launches as ONE graph launch (VBX_RP_GRAPH=0: plain launches): the executable graph and the arguments it was built for
  hipGraph_t rp_graph = nullptr;
  hipGraphExec_t rp_graph_exec = nullptr;
  std::vector<unsigned char> rp_graph_key;
  std::vector<uint32_t> h_midx, h_mhash, h_morder, h_mseq, h_mruns;
  std::vector<int32_t> h_mnxt;
  std::vector<uint32_t> h_poff;  // scratch of the blocked observed-set replay
  uint32_t h_poff_total = 0;     // probes of the current replay guess
  uint32_t rp_last_p = 0;        // probes of the last replay round of the previous frame (sizes the first batch)
  // voxel_observed_approx_set_ (reference semantics, fast_observed_set == 0)
  uint32_t obsset_offset = 0;
  bool obsset_sentinel_live = true;
  bool obsset_init = false;
  // (FastTsdfIntegrator's reset counter is a function-static in the reference, shared by every instance of the process:
  // g_fast_reset_counter below, not a member)
  DBuf b_own0, b_own1;
  // ESDF layer (allocated on first use)
  DBuf b_edist, b_estate, b_eraised, b_eactive;
  // parallel reference-order open set (vbx_kernels_esdf_replay.hpp): control block, records, targets, lists
  DBuf rp_ctl, rp_nbslot, rp_chunk_tab, rp_rec_u32, rp_rec_T, rp_rec_kid, rp_rec_tgts, rp_rec_push, rp_vox2tgt, rp_tgt_u32, rp_tgt_ev, rp_dl,
      rp_lists, rp_sub, rp_sub_list, rp_sim_q, rp_ord, rp_wg_stats, rp_scan_desc, rp_hazard, cls_pos, cls_nb27, cls_shadow, cls_counters;
  size_t rp_vox2tgt_zeroed = 0;   // bytes of rp_vox2tgt known to be zero
  uint32_t rp_rec_cap = 0, rp_tgt_cap = 0, rp_kmax = 0, rp_smax = 0, rp_scan_tiles_cap = 0;
  // Blocks the reference-order updates of this ESDF layer have walked so far, and whether the update in hand walks more than
  // all of them together (the first update of a map): such an update's few floods run for hundreds of rings
  // and are cheaper cut at 64 iterations and continued from the queues, while the 70-ring floods of an incremental update
  // want their 128 (first update of the configs[3] stream 405 -> 398 ms, later ones 25.0 against 25.7 ms at 64).
  size_t rp_walked_total = 0;
  bool rp_bulk = false;
  bool esdf_init = false;
  bool esdf_robot_pending = false;  // addNewRobotPosition since the last update
  // addNewRobotPosition under reference_order: what the call left in the integrator's containers, in the reference's order
  // (raise_ and open_ are members that survive the call, esdf_integrator.cc:48, :84; updated_blocks_ :54, :80)
  bool esdf_robot_ordered = false;              // the pending work is in these lists (not in the order-free marks)
  int esdf_robot_buckets = 0;                   // num_buckets the open_ entries were binned with
  float esdf_robot_max_distance = 0.0f;
  std::vector<uint32_t> esdf_seed_raise;        // pool voxel ids, push order
  std::vector<uint32_t> esdf_seed_open;         // pool voxel ids, push order
  std::vector<uint8_t> esdf_seed_open_bucket;   // bucket of every open_ entry (the distance at push time decides)
  std::vector<HostBlockIdx> esdf_updated_seq;   // updated_blocks_: first insertions, in sequence
  std::unordered_set<HostBlockIdx, HostAnyIndexHash> esdf_updated_set;   // the same as the reference's IndexSet (iteration order)
  void esdf_robot_forget() {   // EsdfIntegrator::clear() (esdf_integrator.h), or the voxels the entries name are gone
    esdf_robot_pending = false;
    esdf_robot_ordered = false;
    esdf_seed_raise.clear();
    esdf_seed_open.clear();
    esdf_seed_open_bucket.clear();
    esdf_updated_seq.clear();
    esdf_updated_set.clear();
  }
  int esdf_spec_raise = 0, esdf_spec_lower = 0;  // sweeps queued ahead of the read-back (esdf_update_t)
  // mesher output of the last vbx_mesh_generate call (vbx_host_mesh.hpp)
  DBuf b_mesh_list, b_mesh_cnt, b_mesh_off, b_mesh_tab, b_mesh_verts, b_mesh_normals, b_mesh_colors;
  std::vector<int32_t> mesh_idx;        // block index per meshed block
  std::vector<uint32_t> mesh_off{0u};   // triangle offsets, one more than blocks
  bool mesh_has_colors = false;
  uint32_t own_tag = 0;  // descending
  int own_s_bits = 0;

  // per-kernel profile (vbx_profile_enable): every launch of the hot path bracketed by two events
  struct ProfRec { const char* name; hipEvent_t e0, e1; };
  bool prof = false;
  std::vector<ProfRec> prof_recs;             // launches of the call in flight
  std::vector<hipEvent_t> prof_pool;          // recycled events
  std::map<std::string, std::pair<uint64_t, double>> prof_table;  // name -> (launches, total ms)
  uint64_t prof_calls = 0;

  vbx_counters counters{};
  bool timing = false;
  hipEvent_t ev[9] = {};  // 0..7 stage boundaries in order, 8 = end of the exact-set solve inside stage 3
  bool ev_hit[9] = {};
  vbx_timing last_timing{};

  void fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    err = buf;
  }
};


// Per-kernel profile: KLAUNCH is hipLaunchKernelGGL plus, when the handle's profile is on, an event
// before and after the launch on the launch stream; prof_collect (end of every API call) turns the
// pairs into per-kernel launch counts and durations.  Off (the default) it costs one branch.
inline hipEvent_t prof_event(vbx_ctx* ctx) {
  if (!ctx->prof_pool.empty()) {
    hipEvent_t e = ctx->prof_pool.back();
    ctx->prof_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
inline void prof_begin(vbx_ctx* ctx, const char* name) {
  if (!ctx->prof) return;
  vbx_ctx::ProfRec r{name, prof_event(ctx), prof_event(ctx)};
  (void)hipEventRecord(r.e0, ctx->stream);
  ctx->prof_recs.push_back(r);
}
inline void prof_end(vbx_ctx* ctx) {
  if (!ctx->prof || ctx->prof_recs.empty()) return;
  (void)hipEventRecord(ctx->prof_recs.back().e1, ctx->stream);
}
inline void prof_collect(vbx_ctx* ctx) {
  if (ctx->prof_recs.empty()) return;
  (void)hipStreamSynchronize(ctx->stream);
  for (const auto& r : ctx->prof_recs) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
      std::string n(r.name);
      if (!n.empty() && n.front() == '(' && n.back() == ')') n = n.substr(1, n.size() - 2);
      auto& t = ctx->prof_table[n];
      t.first += 1;
      t.second += ms;
    }
    ctx->prof_pool.push_back(r.e0);
    ctx->prof_pool.push_back(r.e1);
  }
  ctx->prof_recs.clear();
  ++ctx->prof_calls;
}
#define KLAUNCH(kern, ...)                      \
  do {                                          \
    prof_begin(ctx, #kern);                     \
    hipLaunchKernelGGL(kern, __VA_ARGS__);      \
    prof_end(ctx);                              \
  } while (0)
