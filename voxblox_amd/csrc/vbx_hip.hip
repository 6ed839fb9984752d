// libvbx_hip.so — MI355X (gfx950) TSDF integration hot path of voxblox, hand-written HIP.
//
// What this translation unit implements (reference = /root/reference/voxblox):
//   {Simple,Merged,Fast}TsdfIntegrator::integratePointCloud  src/integrator/tsdf_integrator.cc:242-590
//   RayCaster / ThreadSafeIndex                               src/integrator/integrator_utils.cc
//   Layer<TsdfVoxel> / Block<TsdfVoxel> storage               include/voxblox/core/{layer,block}.h
//   EsdfIntegrator (update, robot spheres)                    src/integrator/esdf_integrator.cc
// behind the C-ABI of include/vbx_hip.h (implemented at the bottom of this file).  The kernels
// and their host orchestration live in the vbx_*.hpp pieces included below; DESIGN.md has the
// data layout and the kernel list.
//
// Design in one paragraph.  The map lives in HBM as a struct-of-arrays block pool
// (dist[], weight[], rgba[] — vps^3 voxels per block, so a block is three contiguous
// 16 KiB / 16 KiB / 16 KiB runs at vps=16) addressed through an open-addressing hash map
// BlockIndex -> pool slot.  updateTsdfVoxel clamps after every update, so the per-voxel fold
// is neither associative nor commutative: the reference's 1-thread result is
// reproduced by marching every ray once to EMIT (voxel, order) keys, sorting them, and folding
// each voxel's updates sequentially in the reference's point order — parallel across voxels,
// ordered within a voxel.  No MFMA anywhere: this is a raycast/scatter path.
//
// Built with -ffp-contract=off (see vbx_device_math.hpp for why).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <map>
#include <memory_resource>
#include <mutex>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>


#include "../../include/vbx_hip.h"
#include "vbx_device_math.hpp"

using namespace vbx;

// The pieces of this translation unit, in dependency order:
#include "vbx_common.hpp"
#include "vbx_kernels_map.hpp"
#include "vbx_kernels_tsdf.hpp"
#include "vbx_kernels_fast.hpp"
#include "vbx_kernels_esdf.hpp"
#include "vbx_kernels_esdf_replay.hpp"
#include "vbx_kernels_esdf_strict.hpp"
#include "vbx_kernels_esdf_classify.hpp"
#include "vbx_kernels_mesh.hpp"
#include "vbx_ctx.hpp"
#include "vbx_sort.hpp"
#include "vbx_host_common.hpp"
#include "vbx_host_tsdf.hpp"
#include "vbx_host_esdf.hpp"
#include "vbx_host_mesh.hpp"

// ---------------------------------------------------------------------------
// C-ABI
// ---------------------------------------------------------------------------
extern "C" {

void vbx_tsdf_cfg_default(vbx_tsdf_cfg* c) {  // tsdf_integrator.h:59-86
  c->default_truncation_distance = 0.1f;
  c->max_weight = 10000.0f;
  c->voxel_carving_enabled = 1;
  c->min_ray_length_m = 0.1f;
  c->max_ray_length_m = 5.0f;
  c->use_const_weight = 0;
  c->allow_clear = 1;
  c->use_weight_dropoff = 1;
  c->use_sparsity_compensation_factor = 0;
  c->sparsity_compensation_factor = 1.0f;
  c->integrator_threads = 1;
  c->integration_order_mode = 0;
  c->enable_anti_grazing = 0;
  c->start_voxel_subsampling_factor = 2.0f;
  c->max_consecutive_ray_collisions = 2;
  c->clear_checks_every_n_frames = 1;
  c->max_integration_time_s = 3.402823466e+38f;
  c->merged_bundle_order = 0;
  c->fast_observed_set = 0;
}

void vbx_esdf_cfg_default(vbx_esdf_cfg* c) {  // esdf_integrator.h:37-77
  c->full_euclidean_distance = 0;
  c->max_distance_m = 2.0f;
  c->min_distance_m = 0.2f;
  c->default_distance_m = 2.0f;
  c->min_diff_m = 0.001f;
  c->min_weight = 1e-6f;
  c->num_buckets = 20;
  c->multi_queue = 0;
  c->add_occupied_crust = 0;
  c->clear_sphere_radius = 1.5f;
  c->occupied_sphere_radius = 5.0f;
  c->reference_order = 1;   // the reference's own result by default (round 5); 0 = the order-free fast mode
}

const char* vbx_last_error(vbx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

vbx_ctx* vbx_create(const vbx_map_cfg* cfg, int device) {
  auto bail = [](vbx_ctx* c, const std::string& msg) -> vbx_ctx* {
    g_create_error = msg;
    if (c) vbx_destroy(c);
    return nullptr;
  };
  if (!cfg || !(cfg->voxel_size > 0.0f)) return bail(nullptr, "vbx_create: voxel_size must be > 0");
  const uint32_t vps = cfg->voxels_per_side;
  if (vps < 4 || vps > 32 || (vps & (vps - 1))) return bail(nullptr, "vbx_create: voxels_per_side must be a power of two in [4,32]");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return bail(nullptr, "vbx_create: no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return bail(nullptr, "vbx_create: bad device ordinal");
  if (hipSetDevice(device) != hipSuccess) return bail(nullptr, "vbx_create: hipSetDevice failed");

  vbx_ctx* ctx = new vbx_ctx;
  ctx->device = device;
  ctx->mcfg = *cfg;
  if (ctx->mcfg.max_blocks == 0) ctx->mcfg.max_blocks = 65536;
  MapDev& m = ctx->map;
  m.cap_blocks = ctx->mcfg.max_blocks;
  m.vps = (int)vps;
  m.vps_log2 = (int)bits_for(vps) - 1;
  m.nvox = vps * vps * vps;
  if ((uint64_t)m.cap_blocks * m.nvox >= (1ull << 32)) {
    return bail(ctx, "vbx_create: max_blocks * voxels_per_side^3 must be < 2^32");
  }
  // Derived constants exactly as Layer's constructor / TsdfIntegratorBase::setLayer compute
  // them (layer.h:34-44, tsdf_integrator.cc:73-79): double division, rounded to float.
  m.voxel_size = cfg->voxel_size;
  m.voxel_size_inv = (float)(1.0 / (double)cfg->voxel_size);
  m.vps_inv = (float)(1.0 / (double)vps);
  uint32_t hcap = 1;
  while (hcap < 4u * m.cap_blocks) hcap <<= 1;
  ctx->hcap = hcap;
  m.hmask = hcap - 1;

  auto ok = [&](hipError_t e) { return e == hipSuccess; };
  const size_t nv = (size_t)m.cap_blocks * m.nvox;
  if (!ok(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)))
    return bail(ctx, "vbx_create: hipStreamCreate failed");
  ctx->stream = ctx->own_stream;
  if (!ok(ctx->b_hkeys.ensure((size_t)hcap * 8)) || !ok(ctx->b_hvals.ensure((size_t)hcap * 4)) ||
      !ok(ctx->b_dist.ensure(nv * 4)) || !ok(ctx->b_weight.ensure(nv * 4)) ||
      !ok(ctx->b_rgba.ensure(nv * 4)) || !ok(ctx->b_blkidx.ensure((size_t)m.cap_blocks * 12)) ||
      !ok(ctx->b_blkflags.ensure((size_t)m.cap_blocks * 4)) || !ok(ctx->b_blkfirst.ensure((size_t)m.cap_blocks * 8)) ||
      !ok(ctx->b_freelist.ensure((size_t)m.cap_blocks * 4)) ||
      !ok(ctx->b_newlist.ensure((size_t)m.cap_blocks * 4)) ||
      !ok(hipMalloc((void**)&ctx->d_state, sizeof(DevState))))
    return bail(ctx, "vbx_create: out of device memory for the block pool");
  {  // host-visible state mirror (optional: VBX_NO_MIRROR=1 or a failed allocation fall back to copies)
    void* hp = nullptr;
    void* dp = nullptr;
    if (!getenv("VBX_NO_MIRROR") &&
        hipHostMalloc(&hp, sizeof(StateMirror), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
      if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
        std::memset(hp, 0, sizeof(StateMirror));
        ctx->h_mirror = static_cast<StateMirror*>(hp);
        ctx->d_mirror = static_cast<StateMirror*>(dp);
      } else {
        (void)hipHostFree(hp);
      }
    }
    (void)hipGetLastError();
  }
  m.hkeys = ctx->b_hkeys.as<uint64_t>();
  m.hvals = ctx->b_hvals.as<uint32_t>();
  m.dist = ctx->b_dist.as<float>();
  m.weight = ctx->b_weight.as<float>();
  m.rgba = ctx->b_rgba.as<uint32_t>();
  m.blk_idx = ctx->b_blkidx.as<int32_t>();
  m.blk_flags = ctx->b_blkflags.as<uint32_t>();
  m.blk_first = ctx->b_blkfirst.as<unsigned long long>();
  m.free_list = ctx->b_freelist.as<uint32_t>();
  hipStream_t s = ctx->stream;
  bool good = ok(hipMemsetAsync(m.hkeys, 0xFF, (size_t)hcap * 8, s)) &&
              ok(hipMemsetAsync(m.hvals, 0xFF, (size_t)hcap * 4, s)) &&
              ok(hipMemsetAsync(m.dist, 0, nv * 4, s)) && ok(hipMemsetAsync(m.weight, 0, nv * 4, s)) &&
              ok(hipMemsetAsync(m.rgba, 0, nv * 4, s)) &&
              ok(hipMemsetAsync(m.blk_flags, 0, (size_t)m.cap_blocks * 4, s)) &&
              ok(hipMemsetAsync(m.blk_first, 0xFF, (size_t)m.cap_blocks * 8, s)) &&
              ok(hipMemsetAsync(ctx->d_state, 0, sizeof(DevState), s));
  for (int i = 0; i < 9 && good; ++i) good = ok(hipEventCreate(&ctx->ev[i]));
  good = good && ok(hipEventCreateWithFlags(&ctx->ev_copy, hipEventDisableTiming));
  good = good && ok(hipStreamSynchronize(s));
  if (!good) return bail(ctx, "vbx_create: device initialisation failed");
  return ctx;
}

void vbx_destroy(vbx_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
  DBuf* bufs[] = {&ctx->b_hkeys, &ctx->b_hvals, &ctx->b_dist, &ctx->b_weight, &ctx->b_rgba,
                  &ctx->b_blkidx, &ctx->b_blkflags, &ctx->b_blkfirst, &ctx->b_newlog, &ctx->b_freelist, &ctx->b_newlist, &ctx->b_pts,
                  &ctx->b_cols, &ctx->t_px, &ctx->t_py, &ctx->t_pz, &ctx->t_rgba, &ctx->t_w,
                  &ctx->t_flags, &ctx->t_bkey, &ctx->u_px, &ctx->u_py, &ctx->u_pz, &ctx->u_rgba,
                  &ctx->u_w, &ctx->u_flags, &ctx->u_bkey, &ctx->b_pcx, &ctx->b_pcy, &ctx->b_pcz,
                  &ctx->b_cnt, &ctx->b_off, &ctx->b_keys0, &ctx->b_keys1, &ctx->b_vals0,
                  &ctx->b_vals1, &ctx->b_tmp, &ctx->b_head, &ctx->b_rank, &ctx->b_graze, &ctx->b_T,
                  &ctx->b_U, &ctx->b_vox, &ctx->b_vhash, &ctx->b_TH, &ctx->b_cl, &ctx->b_act0, &ctx->b_act1, &ctx->b_long, &ctx->b_order, &ctx->b_obs, &ctx->b_sphere0, &ctx->b_sphere1, &ctx->b_redo, &ctx->b_fix, &ctx->b_bkeys, &ctx->b_bfirst, &ctx->b_bperm, &ctx->b_obsset, &ctx->b_collided, &ctx->b_hist0, &ctx->b_hist1, &ctx->b_moved, &ctx->b_scan_desc, &ctx->b_fs_hist, &ctx->b_fs_desc, &ctx->b_bbox, &ctx->w_px, &ctx->w_py, &ctx->w_pz, &ctx->w_rgba, &ctx->w_w, &ctx->w_flags, &ctx->w_bkey, &ctx->b_startset, &ctx->b_own0, &ctx->b_own1, &ctx->b_edist, &ctx->b_estate,
                  &ctx->b_eraised, &ctx->b_eactive, &ctx->b_mesh_list, &ctx->b_mesh_cnt, &ctx->b_mesh_off,
                  &ctx->b_mesh_tab, &ctx->b_mesh_verts, &ctx->b_mesh_normals, &ctx->b_mesh_colors, &ctx->b_bstart, &ctx->b_mgather, &ctx->b_fin};
  for (DBuf* b : bufs) b->release();
  DBuf* rp_bufs[] = {&ctx->rp_ctl, &ctx->rp_nbslot, &ctx->rp_chunk_tab, &ctx->rp_rec_u32, &ctx->rp_rec_T, &ctx->rp_rec_kid,
                     &ctx->rp_rec_tgts, &ctx->rp_rec_push, &ctx->rp_vox2tgt, &ctx->rp_tgt_u32, &ctx->rp_tgt_ev, &ctx->rp_dl,
                     &ctx->rp_lists, &ctx->rp_sub, &ctx->rp_sub_list, &ctx->rp_sim_q, &ctx->rp_ord, &ctx->rp_scan_desc,
                     &ctx->rp_hazard, &ctx->cls_pos, &ctx->cls_nb27, &ctx->cls_shadow, &ctx->cls_counters, &ctx->rp_wg_stats};
  for (DBuf* b : rp_bufs) b->release();
  ctx->rp_h_done.release();
  if (ctx->rp_graph_exec) (void)hipGraphExecDestroy(ctx->rp_graph_exec);
  if (ctx->rp_graph) (void)hipGraphDestroy(ctx->rp_graph);
  for (hipEvent_t e : ctx->rp_look_ev)
    if (e) (void)hipEventDestroy(e);
  ctx->h_mkeys.release();
  ctx->h_mperm.release();
  if (ctx->d_state) (void)hipFree(ctx->d_state);
  if (ctx->h_mirror) (void)hipHostFree(ctx->h_mirror);
  for (int i = 0; i < 9; ++i)
    if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
  if (ctx->ev_copy) (void)hipEventDestroy(ctx->ev_copy);
  for (hipEvent_t e : ctx->prof_pool) (void)hipEventDestroy(e);
  for (const auto& r : ctx->prof_recs) {
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

int vbx_get_map_cfg(vbx_ctx* ctx, vbx_map_cfg* out) {
  if (!ctx || !out) return VBX_ERR_INVALID;
  out->voxel_size = ctx->map.voxel_size;
  out->voxels_per_side = (uint32_t)ctx->map.vps;
  out->max_blocks = ctx->map.cap_blocks;
  return VBX_OK;
}

int vbx_set_stream(vbx_ctx* ctx, void* hip_stream) {
  if (!ctx) return VBX_ERR_INVALID;
  ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
  return VBX_OK;
}

int vbx_tsdf_integrate_device(vbx_ctx* ctx, int kind, const vbx_tsdf_cfg* cfg, const float pos[3],
                              const float quat[4], const float* d_points_C, const uint8_t* d_rgba,
                              size_t n, int freespace_points) {
  if (!ctx) return VBX_ERR_INVALID;
  const int rc = integrate_device(ctx, kind, cfg, pos, quat, d_points_C, d_rgba, n, freespace_points);
  prof_collect(ctx);
  return rc;
}

int vbx_tsdf_integrate(vbx_ctx* ctx, int kind, const vbx_tsdf_cfg* cfg, const float pos[3],
                       const float quat[4], const float* points_C, const uint8_t* rgba, size_t n,
                       int freespace_points) {
  if (!ctx) return VBX_ERR_INVALID;
  if (n && (!points_C || !rgba)) {
    ctx->fail("vbx_tsdf_integrate: null input");
    return VBX_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  if (n) {
    HIP_TRY(ctx->b_pts.ensure(n * 12));
    HIP_TRY(ctx->b_cols.ensure(n * 4));
    HIP_TRY(hipMemcpyAsync(ctx->b_pts.p, points_C, n * 12, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->b_cols.p, rgba, n * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  const int rc = integrate_device(ctx, kind, cfg, pos, quat, ctx->b_pts.as<float>(),
                                  ctx->b_cols.as<uint8_t>(), n, freespace_points);
  prof_collect(ctx);
  return rc;
}

int vbx_esdf_update(vbx_ctx* ctx, const vbx_esdf_cfg* cfg, int batch, int clear_updated_flag) {
  if (!ctx || !cfg) return VBX_ERR_INVALID;
  const int rc = esdf_update(ctx, cfg, batch, clear_updated_flag);
  prof_collect(ctx);
  return rc;
}

int vbx_esdf_reserve(vbx_ctx* ctx, const vbx_esdf_cfg* cfg) {
  if (!ctx || !cfg) return VBX_ERR_INVALID;
  return esdf_reserve(ctx, cfg);
}

int vbx_esdf_update_blocks(vbx_ctx* ctx, const vbx_esdf_cfg* cfg, const int32_t* idx_xyz, size_t n, int incremental) {
  if (!ctx || !cfg || (n && !idx_xyz)) return VBX_ERR_INVALID;
  static const int32_t kNone[3] = {0, 0, 0};
  return esdf_update(ctx, cfg, 0, 0, n ? idx_xyz : kNone, n, incremental);
}

int vbx_esdf_integrator_clear(vbx_ctx* ctx) {
  if (!ctx) return VBX_ERR_INVALID;
  HIP_TRY(hipSetDevice(ctx->device));
  ctx->esdf_robot_forget();
  if (!ctx->esdf_init) return VBX_OK;
  int rc = sync_state(ctx);
  if (rc) return rc;
  const uint32_t used = ctx->h_state.pool_used;
  if (used == 0) return VBX_OK;
  HIP_TRY(hipMemsetAsync(ctx->b_eraised.p, 0, (size_t)used * ctx->map.nvox, ctx->stream));
  hipLaunchKernelGGL(k_clear_update_bits, grid_for(used), dim3(256), 0, ctx->stream, ctx->map, used, ~0u,
                     kFlagEsdfPendClassify | kFlagEsdfPendOpen);
  return VBX_OK;
}

int64_t vbx_selftest_index_set_order(const int32_t* idx_xyz, size_t n, int32_t* out_xyz) {
  if ((n && !idx_xyz) || !out_xyz) return -1;
  std::unordered_set<HostBlockIdx, HostAnyIndexHash> set;
  for (size_t i = 0; i < n; ++i) set.insert(HostBlockIdx{idx_xyz[3 * i], idx_xyz[3 * i + 1], idx_xyz[3 * i + 2]});
  size_t k = 0;
  for (const HostBlockIdx& b : set) {
    out_xyz[3 * k] = b.x; out_xyz[3 * k + 1] = b.y; out_xyz[3 * k + 2] = b.z;
    ++k;
  }
  return (int64_t)k;
}

int vbx_esdf_robot_updated_blocks(vbx_ctx* ctx, int order, int32_t* out_xyz, size_t cap, size_t* n_out, int clear) {
  if (!ctx || !n_out || (order != 0 && order != 1)) return VBX_ERR_INVALID;
  *n_out = ctx->esdf_updated_set.size();
  if (out_xyz) {
    size_t k = 0;
    auto put = [&](const HostBlockIdx& b) {
      if (k < cap) { out_xyz[3 * k] = b.x; out_xyz[3 * k + 1] = b.y; out_xyz[3 * k + 2] = b.z; }
      ++k;
    };
    if (order == 0) for (const HostBlockIdx& b : ctx->esdf_updated_seq) put(b);
    else for (const HostBlockIdx& b : ctx->esdf_updated_set) put(b);
  }
  if (clear) {
    ctx->esdf_updated_set.clear();
    ctx->esdf_updated_seq.clear();
  }
  return VBX_OK;
}

int vbx_esdf_add_new_robot_position(vbx_ctx* ctx, const vbx_esdf_cfg* cfg, const float position[3]) {
  if (!ctx || !cfg || !position) return VBX_ERR_INVALID;
  return esdf_add_new_robot_position(ctx, cfg, position);
}

// ---- block listing / transfer --------------------------------------------------------
static int list_blocks(vbx_ctx* ctx, int layer, uint32_t need_mask, std::vector<std::pair<uint64_t, uint32_t>>* out) {
  if (layer != VBX_LAYER_TSDF && layer != VBX_LAYER_ESDF) {
    ctx->fail("unknown layer %d", layer);
    return VBX_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  int rc = sync_state(ctx);
  if (rc) return rc;
  const uint32_t used = ctx->h_state.pool_used;
  std::vector<uint32_t> flags(used);
  std::vector<int32_t> idx((size_t)used * 3);
  if (used) {
    HIP_TRY(hipMemcpy(flags.data(), ctx->map.blk_flags, (size_t)used * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(idx.data(), ctx->map.blk_idx, (size_t)used * 12, hipMemcpyDeviceToHost));
  }
  out->clear();
  for (uint32_t sl = 0; sl < used; ++sl) {
    if (layer == VBX_LAYER_ESDF) {
      if (!(flags[sl] & kFlagEsdfAlloc)) continue;
      // VBX_UPDATE_DIRTY (8): voxels changed since the mirror last took the block — a device-side bit next to the
      // block's Update bits (the wavefront writes blocks the reference never flags as updated)
      const uint32_t have = ((flags[sl] >> kFlagEsdfUpdShift) & kFlagUpdMask) | ((flags[sl] & kFlagEsdfDirty) ? 8u : 0u);
      if (need_mask && !(have & need_mask)) continue;
    } else {
      if (!(flags[sl] & kFlagPublished)) continue;
      if (need_mask && !(flags[sl] & need_mask)) continue;
    }
    out->emplace_back(pack_block_key(idx[3 * sl], idx[3 * sl + 1], idx[3 * sl + 2]), sl);
  }
  std::sort(out->begin(), out->end());
  return VBX_OK;
}

int vbx_num_blocks(vbx_ctx* ctx, int layer, size_t* n) {
  if (!ctx || !n) return VBX_ERR_INVALID;
  std::vector<std::pair<uint64_t, uint32_t>> v;
  int rc = list_blocks(ctx, layer, 0, &v);
  if (rc) return rc;
  *n = v.size();
  return VBX_OK;
}

static int emit_list(const std::vector<std::pair<uint64_t, uint32_t>>& v, int32_t* idx, size_t cap, size_t* n) {
  *n = v.size();
  for (size_t i = 0; i < v.size() && i < cap; ++i) {
    int x, y, z;
    unpack_block_key(v[i].first, &x, &y, &z);
    idx[3 * i] = x; idx[3 * i + 1] = y; idx[3 * i + 2] = z;
  }
  return VBX_OK;
}

int vbx_block_indices(vbx_ctx* ctx, int layer, int32_t* idx, size_t cap, size_t* n) {
  if (!ctx || !n || (cap && !idx)) return VBX_ERR_INVALID;
  std::vector<std::pair<uint64_t, uint32_t>> v;
  int rc = list_blocks(ctx, layer, 0, &v);
  if (rc) return rc;
  return emit_list(v, idx, cap, n);
}

int vbx_blocks_updated(vbx_ctx* ctx, int layer, int update_mask, int32_t* idx, size_t cap, size_t* n) {
  if (!ctx || !n || (cap && !idx)) return VBX_ERR_INVALID;
  std::vector<std::pair<uint64_t, uint32_t>> v;
  int rc = list_blocks(ctx, layer, (uint32_t)update_mask & (kFlagUpdMask | 8u), &v);
  if (rc) return rc;
  return emit_list(v, idx, cap, n);
}

int vbx_blocks_new_ordered(vbx_ctx* ctx, int32_t* idx, size_t cap, size_t* n) {
  if (!ctx || !n || (cap && !idx)) return VBX_ERR_INVALID;
  HIP_TRY(hipSetDevice(ctx->device));
  int rc = drain_new_blocks(ctx);
  if (rc) return rc;
  if (ctx->last_new_seq != ctx->last_call_seq) ctx->last_new.clear();   // the last call published nothing
  *n = ctx->last_new.size();
  for (size_t i = 0; i < ctx->last_new.size() && i < cap; ++i) {
    idx[3 * i] = ctx->last_new[i].x; idx[3 * i + 1] = ctx->last_new[i].y; idx[3 * i + 2] = ctx->last_new[i].z;
  }
  return VBX_OK;
}

int vbx_set_block_order_tracking(vbx_ctx* ctx, int on) {
  if (!ctx) return VBX_ERR_INVALID;
  HIP_TRY(hipSetDevice(ctx->device));
  int rc = sync_state(ctx);
  if (rc) return rc;
  if (ctx->new_flags_live) {
    rc = collect_new_blocks(ctx);
    if (rc) return rc;
  }
  rc = discard_new_blocks(ctx);   // (what the library knows about the Layer's order ends here)
  if (rc) return rc;
  ctx->track_block_order = on != 0;
  ctx->layer_order_exact = ctx->track_block_order && ctx->h_state.pool_used == 0;   // blocks already in the map were not followed
  ctx->map.blk_first = ctx->track_block_order ? ctx->b_blkfirst.as<unsigned long long>() : nullptr;
  return VBX_OK;
}

int vbx_block_indices_layer_order(vbx_ctx* ctx, int update_mask, int32_t* idx, size_t cap, size_t* n, int* exact) {
  if (!ctx || !n || (cap && !idx)) return VBX_ERR_INVALID;
  std::vector<std::pair<uint64_t, uint32_t>> v;
  int rc = list_blocks(ctx, VBX_LAYER_TSDF, (uint32_t)update_mask & kFlagUpdMask, &v);
  if (rc) return rc;
  bool ex = true;
  rc = order_like_layer(ctx, &v, &ex);
  if (rc) return rc;
  if (exact) *exact = ex ? 1 : 0;
  return emit_list(v, idx, cap, n);
}

static int find_slot_host(vbx_ctx* ctx, const int32_t idx[3], uint32_t* slot, uint32_t* hpos) {
  // host-side probe of the device hash map (small D2H reads; not on the hot path)
  const uint64_t key = pack_block_key(idx[0], idx[1], idx[2]);
  uint32_t h = mix_key(key) & ctx->map.hmask;
  for (uint32_t probes = 0; probes <= ctx->map.hmask; ++probes) {
    uint64_t k;
    HIP_TRY(hipMemcpy(&k, ctx->map.hkeys + h, 8, hipMemcpyDeviceToHost));
    if (k == key) {
      HIP_TRY(hipMemcpy(slot, ctx->map.hvals + h, 4, hipMemcpyDeviceToHost));
      if (hpos) *hpos = h;
      return VBX_OK;
    }
    if (k == kEmptyKey) break;
    h = (h + 1) & ctx->map.hmask;
  }
  *slot = kInvalidSlot;
  return VBX_OK;
}

int vbx_block_download(vbx_ctx* ctx, int layer, const int32_t idx[3], void* aos, uint8_t* updated_bits,
                       uint8_t* has_data) {
  if (!ctx || !idx || !aos) return VBX_ERR_INVALID;
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  uint32_t slot;
  int rc = find_slot_host(ctx, idx, &slot, nullptr);
  if (rc) return rc;
  uint32_t flags = 0;
  if (slot != kInvalidSlot) HIP_TRY(hipMemcpy(&flags, ctx->map.blk_flags + slot, 4, hipMemcpyDeviceToHost));
  const uint32_t need = (layer == VBX_LAYER_ESDF) ? kFlagEsdfAlloc : kFlagPublished;
  if (slot == kInvalidSlot || !(flags & need) || (layer == VBX_LAYER_ESDF && !ctx->esdf_init)) {
    ctx->fail("block (%d,%d,%d) is not allocated", idx[0], idx[1], idx[2]);
    return VBX_ERR_INVALID;
  }
  const uint32_t nv = ctx->map.nvox;
  if (layer == VBX_LAYER_ESDF) {
    std::vector<float> d(nv);
    std::vector<uint32_t> st(nv);
    HIP_TRY(hipMemcpy(d.data(), ctx->b_edist.as<float>() + (size_t)slot * nv, nv * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(st.data(), ctx->b_estate.as<uint32_t>() + (size_t)slot * nv, nv * 4, hipMemcpyDeviceToHost));
    uint8_t* o = static_cast<uint8_t*>(aos);
    for (uint32_t i = 0; i < nv; ++i) {  // EsdfVoxel AoS, voxel.h:18-37 (20 bytes)
      std::memcpy(o + 20 * i, &d[i], 4);
      o[20 * i + 4] = (st[i] & 1) ? 1 : 0;
      o[20 * i + 5] = (st[i] & 2) ? 1 : 0;
      o[20 * i + 6] = (st[i] & 4) ? 1 : 0;
      o[20 * i + 7] = (st[i] & 8) ? 1 : 0;
      const int32_t p[3] = {(int32_t)(int8_t)((st[i] >> 8) & 0xFF), (int32_t)(int8_t)((st[i] >> 16) & 0xFF),
                            (int32_t)(int8_t)((st[i] >> 24) & 0xFF)};
      std::memcpy(o + 20 * i + 8, p, 12);
    }
    if (updated_bits) *updated_bits = (uint8_t)((flags >> kFlagEsdfUpdShift) & kFlagUpdMask);
    if (has_data) *has_data = 0;
    return VBX_OK;
  }
  std::vector<float> d(nv), w(nv);
  std::vector<uint32_t> c(nv);
  HIP_TRY(hipMemcpy(d.data(), ctx->map.dist + (size_t)slot * nv, nv * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(w.data(), ctx->map.weight + (size_t)slot * nv, nv * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(c.data(), ctx->map.rgba + (size_t)slot * nv, nv * 4, hipMemcpyDeviceToHost));
  uint8_t* o = static_cast<uint8_t*>(aos);
  for (uint32_t i = 0; i < nv; ++i) {  // TsdfVoxel AoS, voxel.h:12-16
    std::memcpy(o + 12 * i, &d[i], 4);
    std::memcpy(o + 12 * i + 4, &w[i], 4);
    std::memcpy(o + 12 * i + 8, &c[i], 4);
  }
  if (updated_bits) *updated_bits = (uint8_t)(flags & kFlagUpdMask);
  if (has_data) *has_data = (flags & kFlagHasData) ? 1 : 0;
  return VBX_OK;
}

void* vbx_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}
void vbx_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

static int upload_idx(vbx_ctx* ctx, const int32_t* idx, size_t n);

int vbx_blocks_download(vbx_ctx* ctx, int layer, const int32_t* idx, size_t n, void* aos, uint8_t* updated_bits,
                        uint8_t* has_data) {
  if (!ctx || (n && (!idx || !aos))) return VBX_ERR_INVALID;
  if (layer != VBX_LAYER_TSDF && layer != VBX_LAYER_ESDF) {
    ctx->fail("unknown layer %d", layer);
    return VBX_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  if (n == 0) return VBX_OK;
  if (layer == VBX_LAYER_ESDF && !ctx->esdf_init) {
    ctx->fail("ESDF layer is empty");
    return VBX_ERR_INVALID;
  }
  int rc = upload_idx(ctx, idx, n);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  MapDev& m = ctx->map;
  const size_t wpb = (size_t)m.nvox * (layer == VBX_LAYER_TSDF ? 3 : 5);
  HIP_TRY(ctx->b_keys0.ensure(n * wpb * 4));
  HIP_TRY(ctx->b_vals0.ensure(n * 4));
  hipLaunchKernelGGL(k_lookup_slots_flags, grid_for(n), dim3(256), 0, s, m, ctx->b_head.as<int32_t>(), (uint32_t)n,
                     layer == VBX_LAYER_TSDF ? kFlagPublished : kFlagEsdfAlloc, ctx->b_rank.as<uint32_t>(),
                     ctx->b_vals0.as<uint32_t>());
  if (layer == VBX_LAYER_TSDF)
    hipLaunchKernelGGL(k_pack_tsdf_aos, dim3((unsigned)n), dim3(256), 0, s, m, ctx->b_rank.as<uint32_t>(),
                       ctx->b_keys0.as<uint32_t>());
  else
    hipLaunchKernelGGL(k_pack_esdf_aos, dim3((unsigned)n), dim3(256), 0, s, m.nvox, ctx->b_edist.as<float>(),
                       ctx->b_estate.as<uint32_t>(), ctx->b_rank.as<uint32_t>(), ctx->b_keys0.as<uint32_t>());
  std::vector<uint32_t> flags(n);
  HIP_TRY(hipMemcpyAsync(flags.data(), ctx->b_vals0.p, n * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(aos, ctx->b_keys0.p, n * wpb * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (size_t i = 0; i < n; ++i) {
    if (flags[i] == ~0u) {
      ctx->fail("block (%d,%d,%d) is not allocated", idx[3 * i], idx[3 * i + 1], idx[3 * i + 2]);
      return VBX_ERR_INVALID;
    }
    if (layer == VBX_LAYER_TSDF) {
      if (updated_bits) updated_bits[i] = (uint8_t)(flags[i] & kFlagUpdMask);
      if (has_data) has_data[i] = (flags[i] & kFlagHasData) ? 1 : 0;
    } else {
      if (updated_bits) updated_bits[i] = (uint8_t)((flags[i] >> kFlagEsdfUpdShift) & kFlagUpdMask);
      if (has_data) has_data[i] = 0;
    }
  }
  return VBX_OK;
}

// Layer::allocateBlockPtrByIndex for the n BlockIndex rows staged in b_head (upload_idx): keys inserted,
// slots assigned (the pool grows when it runs out), the rows' slots left in b_rank.
static int insert_and_lookup(vbx_ctx* ctx, size_t n) {
  hipStream_t s = ctx->stream;
  MapDev& m = ctx->map;
  for (;;) {
    HIP_TRY(hipMemsetAsync(&ctx->d_state->new_count, 0, 4, s));
    HIP_TRY(hipMemsetAsync(&ctx->d_state->error, 0, 4, s));
    hipLaunchKernelGGL(k_insert_blocks, grid_for(n), dim3(256), 0, s, m, ctx->b_head.as<int32_t>(), (uint32_t)n,
                       ctx->b_newlist.as<uint32_t>(), ctx->d_state);
    hipLaunchKernelGGL(k_assign_slots, grid_for(m.cap_blocks), dim3(256), 0, s, m, ctx->b_newlist.as<uint32_t>(),
                       ctx->d_state);
    hipLaunchKernelGGL(k_commit_alloc, dim3(1), dim3(1), 0, s, m, ctx->d_state);
    int rc = sync_state(ctx);
    if (rc) return rc;
    if (!(ctx->h_state.error & 1u)) break;
    rc = grow_pool(ctx);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_lookup_slots, grid_for(n), dim3(256), 0, s, m, ctx->b_head.as<int32_t>(), (uint32_t)n, 0,
                     ctx->b_rank.as<uint32_t>());
  return VBX_OK;
}

// Layer::allocateBlockPtrByIndex + a voxel copy for n blocks at once (loadMap, tsdfMapCallback,
// tsdf_server.cc:566-578, 639-653): keys inserted and slots assigned on the device, the AoS voxels staged
// with one copy and unpacked by one kernel.
int vbx_blocks_upload(vbx_ctx* ctx, int layer, const int32_t* idx, size_t n, const void* aos,
                      const uint8_t* updated_bits, const uint8_t* has_data) {
  if (!ctx || (n && (!idx || !aos || !updated_bits))) return VBX_ERR_INVALID;
  if (layer != VBX_LAYER_TSDF && layer != VBX_LAYER_ESDF) {
    ctx->fail("unknown layer %d", layer);
    return VBX_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  if (n == 0) return VBX_OK;
  int rc = VBX_OK;
  if (layer == VBX_LAYER_ESDF) {
    rc = esdf_ensure(ctx);
    if (rc) return rc;
  }
  rc = upload_idx(ctx, idx, n);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  MapDev& m = ctx->map;
  const size_t bpb = (size_t)m.nvox * (layer == VBX_LAYER_TSDF ? 12 : 20);
  HIP_TRY(ctx->b_keys0.ensure(n * bpb));
  HIP_TRY(ctx->b_graze.ensure(2 * n));
  HIP_TRY(hipMemcpyAsync(ctx->b_keys0.p, aos, n * bpb, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(ctx->b_graze.p, updated_bits, n, hipMemcpyHostToDevice, s));
  uint8_t* d_hd = nullptr;
  if (has_data && layer == VBX_LAYER_TSDF) {
    d_hd = ctx->b_graze.as<uint8_t>() + n;
    HIP_TRY(hipMemcpyAsync(d_hd, has_data, n, hipMemcpyHostToDevice, s));
  }
  rc = insert_and_lookup(ctx, n);
  if (rc) return rc;
  if (layer == VBX_LAYER_TSDF) {
    hipLaunchKernelGGL(k_unpack_tsdf_aos, dim3((unsigned)n), dim3(256), 0, s, m, ctx->b_rank.as<uint32_t>(),
                       ctx->b_keys0.as<uint32_t>());
    // the ESDF layer's membership of the same slot is untouched (the layers are independent)
    hipLaunchKernelGGL(k_replace_block_flags, grid_for(n), dim3(256), 0, s, m, ctx->b_rank.as<uint32_t>(), (uint32_t)n,
                       kEsdfBits, kFlagPublished, ctx->b_graze.as<uint8_t>(), 0, d_hd);
  } else {
    hipLaunchKernelGGL(k_unpack_esdf_aos, dim3((unsigned)n), dim3(256), 0, s, m.nvox, ctx->b_edist.as<float>(),
                       ctx->b_estate.as<uint32_t>(), ctx->b_rank.as<uint32_t>(), ctx->b_keys0.as<uint32_t>());
    hipLaunchKernelGGL(k_replace_block_flags, grid_for(n), dim3(256), 0, s, m, ctx->b_rank.as<uint32_t>(), (uint32_t)n,
                       ~kEsdfBits, kFlagEsdfAlloc | kFlagEsdfUnsettled, ctx->b_graze.as<uint8_t>(), (int)kFlagEsdfUpdShift,
                       (const uint8_t*)nullptr);
  }
  rc = sync_state(ctx);  // the caller's host buffers are free again
  if (rc) return rc;
  rc = check_state_error(ctx);
  if (rc) return rc;
  if (layer == VBX_LAYER_TSDF) {
    // Layer::allocateBlockPtrByIndex in the caller's sequence (layer.h:133-160): blocks the Layer did not hold join it in list order
    rc = drain_new_blocks(ctx);
    if (rc) return rc;
    for (size_t i = 0; i < n; ++i) ctx->layer_order.emplace(HostBlockIdx{idx[3 * i], idx[3 * i + 1], idx[3 * i + 2]}, 0u);
  }
  return VBX_OK;
}

int vbx_block_upload(vbx_ctx* ctx, int layer, const int32_t idx[3], const void* aos, uint8_t updated_bits,
                     uint8_t has_data) {
  return vbx_blocks_upload(ctx, layer, idx, 1, aos, &updated_bits, &has_data);
}

// After a removal: pool slots whose block belongs to neither layer any more give their hash entry up and go
// on the free list (k_reclaim); a map without any block left is reset to its initial state, and a hash
// table that has collected too many tombstones is rebuilt.
static int reclaim_slots(vbx_ctx* ctx) {
  int rc = sync_state(ctx);
  if (rc) return rc;
  const uint32_t used = ctx->h_state.pool_used;
  if (used == 0) return VBX_OK;
  hipStream_t s = ctx->stream;
  MapDev& m = ctx->map;
  HIP_TRY(hipMemsetAsync(&ctx->d_state->live_slots, 0, 4, s));
  hipLaunchKernelGGL(k_reclaim, grid_for(used), dim3(256), 0, s, m, ctx->d_state);
  if (ctx->b_obs.p)  // per-voxel "observed this epoch" stamps must not outlive the block (Fast, exact set, n-frame sets)
    hipLaunchKernelGGL(k_zero_free_slots_u32, dim3(used), dim3(256), 0, s, m, ctx->b_obs.as<uint32_t>());
  rc = sync_state(ctx);
  if (rc) return rc;
  if (ctx->h_state.live_slots == 0) {
    HIP_TRY(hipMemsetAsync(m.hkeys, 0xFF, (size_t)ctx->hcap * 8, s));
    HIP_TRY(hipMemsetAsync(m.hvals, 0xFF, (size_t)ctx->hcap * 4, s));
    HIP_TRY(hipMemsetAsync(m.blk_flags, 0, (size_t)used * 4, s));
    hipLaunchKernelGGL(k_reset_pool, dim3(1), dim3(1), 0, s, ctx->d_state);
  } else if (ctx->h_state.tomb_count > ctx->hcap / 4) {
    HIP_TRY(hipMemsetAsync(m.hkeys, 0xFF, (size_t)ctx->hcap * 8, s));
    HIP_TRY(hipMemsetAsync(m.hvals, 0xFF, (size_t)ctx->hcap * 4, s));
    hipLaunchKernelGGL(k_rehash, grid_for(used), dim3(256), 0, s, m, ctx->d_state);
  }
  return VBX_OK;
}

// Layer::removeBlock for n blocks (layer.h:160-165): one lookup kernel, one removal kernel (a workgroup per listed
// block), ONE reclaim pass — a host loop over single removals paid two read-backs and a pass over the pool each.
int vbx_blocks_remove(vbx_ctx* ctx, int layer, const int32_t* idx, size_t n) {
  if (!ctx || (n && !idx)) return VBX_ERR_INVALID;
  if (layer != VBX_LAYER_TSDF && layer != VBX_LAYER_ESDF) {
    ctx->fail("unknown layer %d", layer);
    return VBX_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  if (n == 0) return VBX_OK;
  if (layer == VBX_LAYER_ESDF && !ctx->esdf_init) return VBX_OK;  // no ESDF block exists
  int rc = upload_idx(ctx, idx, n);
  if (rc) return rc;
  if (layer == VBX_LAYER_TSDF) {
    rc = drain_new_blocks(ctx);
    if (rc) return rc;
    for (size_t i = 0; i < n; ++i) ctx->layer_order.erase(HostBlockIdx{idx[3 * i], idx[3 * i + 1], idx[3 * i + 2]});
  }
  hipStream_t s = ctx->stream;
  hipLaunchKernelGGL(k_lookup_slots, grid_for(n), dim3(256), 0, s, ctx->map, ctx->b_head.as<int32_t>(), (uint32_t)n, 0,
                     ctx->b_rank.as<uint32_t>());
  hipLaunchKernelGGL(k_remove_listed, dim3((unsigned)n), dim3(256), 0, s, ctx->map,
                     ctx->esdf_init ? ctx->b_edist.as<float>() : (float*)nullptr,
                     ctx->esdf_init ? ctx->b_estate.as<uint32_t>() : (uint32_t*)nullptr, layer, ctx->b_rank.as<uint32_t>());
  return reclaim_slots(ctx);  // synchronises: idx is a caller-owned host buffer
}

int vbx_block_remove(vbx_ctx* ctx, int layer, const int32_t idx[3]) {
  if (!ctx || !idx) return VBX_ERR_INVALID;
  return vbx_blocks_remove(ctx, layer, idx, 1);
}

int vbx_remove_distant_blocks(vbx_ctx* ctx, int layer, const float center[3], double max_distance) {
  if (!ctx || !center) return VBX_ERR_INVALID;
  if (layer != VBX_LAYER_TSDF && layer != VBX_LAYER_ESDF) {
    ctx->fail("unknown layer %d", layer);
    return VBX_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  int rc = sync_state(ctx);
  if (rc) return rc;
  if (layer == VBX_LAYER_TSDF) {   // (which blocks go is decided on the device: ordered_block_list drops the stale keys)
    rc = drain_new_blocks(ctx);
    if (rc) return rc;
  }
  const uint32_t used = ctx->h_state.pool_used;
  if (used == 0) return VBX_OK;
  const float block_size = ctx->map.voxel_size * (float)ctx->map.vps;
  hipLaunchKernelGGL(k_remove_distant, dim3(used), dim3(256), 0, ctx->stream, ctx->map,
                     ctx->esdf_init ? ctx->b_edist.as<float>() : (float*)nullptr,
                     ctx->esdf_init ? ctx->b_estate.as<uint32_t>() : (uint32_t*)nullptr, layer,
                     f3{center[0], center[1], center[2]}, max_distance * max_distance, block_size);
  rc = reclaim_slots(ctx);
  if (rc || layer != VBX_LAYER_TSDF) return rc;
  // (a map that does not follow the order, or has no order yet, has nothing to reconcile: tsdf_server-style callers remove
  // distant blocks after every frame)
  if (!ctx->track_block_order || ctx->layer_order.empty()) return VBX_OK;
  // block_map_.erase(it) for every block that went (layer.h:170-182): which ones was decided on the device, so the keys
  // that are not published any more leave the replayed container (an erased key frees its bucket: it matters for where
  // later insertions land)
  std::vector<std::pair<uint64_t, uint32_t>> alive;
  rc = list_blocks(ctx, VBX_LAYER_TSDF, 0, &alive);
  if (rc) return rc;
  std::unordered_set<uint64_t> keys;
  keys.reserve(alive.size() * 2);
  for (const auto& kv : alive) keys.insert(kv.first);
  for (auto it = ctx->layer_order.begin(); it != ctx->layer_order.end();) {
    if (keys.count(pack_block_key(it->first.x, it->first.y, it->first.z))) ++it;
    else it = ctx->layer_order.erase(it);
  }
  return VBX_OK;
}

int vbx_clear(vbx_ctx* ctx, int layer) {
  if (!ctx) return VBX_ERR_INVALID;
  if (layer != VBX_LAYER_TSDF && layer != VBX_LAYER_ESDF) {
    ctx->fail("unknown layer %d", layer);
    return VBX_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  {
    // removeAllBlocks (layer.h:168): every block of the layer is zeroed and leaves it in one
    // launch (a workgroup per pool slot, slots outside the layer exit at once); slots that hold no
    // block of the other layer either are recycled (reclaim_slots).
    int rc = sync_state(ctx);
    if (rc) return rc;
    const uint32_t used = ctx->h_state.pool_used;
    if (layer == VBX_LAYER_ESDF) {
      ctx->esdf_robot_forget();  // (queue entries name voxels of the layer that goes)
      ctx->rp_walked_total = 0;
    }
    if (layer == VBX_LAYER_TSDF) {   // block_map_.clear() (layer.h:168): the keys go, the bucket array stays
      ctx->published_since_clear = 0;
      rc = discard_new_blocks(ctx);
      if (rc) return rc;
    }
    if (used == 0) return VBX_OK;
    hipLaunchKernelGGL(k_remove_distant, dim3(used), dim3(256), 0, ctx->stream, ctx->map,
                       ctx->esdf_init ? ctx->b_edist.as<float>() : (float*)nullptr,
                       ctx->esdf_init ? ctx->b_estate.as<uint32_t>() : (uint32_t*)nullptr, layer, f3{0.f, 0.f, 0.f},
                       -1.0, 0.0f);  // squared distance > -1: every block
    return reclaim_slots(ctx);
  }
}

// removeAllBlocks for a scratch map that will see the same region again (the per-step delta maps of the sharding):
// every TSDF block is zeroed and leaves the layer, but its hash entry and pool slot stay behind as an invisible
// candidate (exactly the state a block has between the moment a ray's path first met it and the moment a ray reaches
// it) — the next frame finds its blocks allocated instead of allocating all of them again and re-walking every ray
// that met a new block.  Falls back to vbx_clear when the map also holds ESDF blocks.
int vbx_clear_keep_slots(vbx_ctx* ctx) {
  if (!ctx) return VBX_ERR_INVALID;
  HIP_TRY(hipSetDevice(ctx->device));
  // (per-voxel "observed this epoch" stamps of the exact-set n-frame mode must not outlive their block either)
  if (ctx->esdf_init || ctx->b_obs.p) return vbx_clear(ctx, VBX_LAYER_TSDF);
  int rc = sync_state(ctx);
  if (rc) return rc;
  const uint32_t used = ctx->h_state.pool_used;
  if (used == 0) return VBX_OK;
  // Keeping the slots keeps every block the map ever touched: a delta map that follows a moving sensor would grow towards the
  // whole map (no reclaim runs here).  Once the pool holds far more than the last call published, the blocks go back for real.
  // (published_since_clear: every integrate call since the last clear — a step may integrate several clouds into one delta map)
  const uint32_t published = ctx->published_since_clear;
  ctx->published_since_clear = 0;
  if (used > 2u * published + 256u) return vbx_clear(ctx, VBX_LAYER_TSDF);
  rc = discard_new_blocks(ctx);
  if (rc) return rc;
  hipLaunchKernelGGL(k_remove_distant, dim3(used), dim3(256), 0, ctx->stream, ctx->map, (float*)nullptr, (uint32_t*)nullptr,
                     VBX_LAYER_TSDF, f3{0.f, 0.f, 0.f}, -1.0, 0.0f);  // squared distance > -1: every block
  return VBX_OK;
}

int vbx_clear_updated(vbx_ctx* ctx, int layer, int update_mask) {
  if (!ctx) return VBX_ERR_INVALID;
  if (layer != VBX_LAYER_TSDF && layer != VBX_LAYER_ESDF) {
    ctx->fail("unknown layer %d", layer);
    return VBX_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  int rc = sync_state(ctx);
  if (rc) return rc;
  const uint32_t used = ctx->h_state.pool_used;
  if (used == 0) return VBX_OK;
  const uint32_t bits = (layer == VBX_LAYER_ESDF) ? ((((uint32_t)update_mask & kFlagUpdMask) << kFlagEsdfUpdShift) |
                                                     (((uint32_t)update_mask & 8u) ? kFlagEsdfDirty : 0u))
                                                  : ((uint32_t)update_mask & kFlagUpdMask);
  hipLaunchKernelGGL(k_clear_update_bits, grid_for(used), dim3(256), 0, ctx->stream, ctx->map, used,
                     layer == VBX_LAYER_ESDF ? kFlagEsdfAlloc : kFlagPublished, bits);
  return VBX_OK;
}


// ---- multi-GPU block merge ------------------------------------------------------------
static int upload_idx(vbx_ctx* ctx, const int32_t* idx, size_t n) {
  HIP_TRY(ctx->b_head.ensure(std::max<size_t>(n, 1) * 12));
  HIP_TRY(ctx->b_rank.ensure(std::max<size_t>(n, 1) * 4));
  HIP_TRY(hipMemcpyAsync(ctx->b_head.p, idx, n * 12, hipMemcpyHostToDevice, ctx->stream));
  return VBX_OK;
}

int vbx_blocks_export_sums(vbx_ctx* ctx, const int32_t* idx, size_t n, float* d_out) {
  if (!ctx || (n && (!idx || !d_out))) return VBX_ERR_INVALID;
  HIP_TRY(hipSetDevice(ctx->device));
  if (n == 0) return VBX_OK;
  int rc = upload_idx(ctx, idx, n);
  if (rc) return rc;
  hipLaunchKernelGGL(k_lookup_slots, grid_for(n), dim3(256), 0, ctx->stream, ctx->map,
                     ctx->b_head.as<int32_t>(), (uint32_t)n, 1, ctx->b_rank.as<uint32_t>());
  hipLaunchKernelGGL(k_export_sums, dim3((unsigned)n), dim3(256), 0, ctx->stream, ctx->map,
                     ctx->b_rank.as<uint32_t>(), d_out);
  HIP_TRY(hipStreamSynchronize(ctx->stream));  // idx is a caller-owned host buffer
  return VBX_OK;
}

int vbx_blocks_merge_sums(vbx_ctx* ctx, const int32_t* idx, size_t n, const float* d_sums, int apply_caps,
                          float truncation_distance, float max_weight) {
  if (!ctx || (n && (!idx || !d_sums))) return VBX_ERR_INVALID;
  HIP_TRY(hipSetDevice(ctx->device));
  if (n == 0) return VBX_OK;
  // rows of the same block (several senders touched it) are summed before the merge, in row order:
  // group the row numbers by key, keys in first-appearance order
  std::vector<uint32_t> order(n);
  for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
  auto key_less = [&](uint32_t a, uint32_t b) {
    const int32_t *p = idx + 3 * (size_t)a, *q = idx + 3 * (size_t)b;
    if (p[2] != q[2]) return p[2] < q[2];
    if (p[1] != q[1]) return p[1] < q[1];
    return p[0] < q[0];
  };
  std::stable_sort(order.begin(), order.end(), key_less);
  std::vector<int32_t> uniq;
  std::vector<uint32_t> row_start;
  for (size_t i = 0; i < n; ++i) {
    if (i == 0 || key_less(order[i - 1], order[i])) {
      row_start.push_back((uint32_t)i);
      uniq.insert(uniq.end(), idx + 3 * (size_t)order[i], idx + 3 * (size_t)order[i] + 3);
    }
  }
  row_start.push_back((uint32_t)n);
  const size_t nu = row_start.size() - 1;
  int rc = upload_idx(ctx, uniq.data(), nu);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  MapDev& m = ctx->map;
  HIP_TRY(ctx->b_vals0.ensure((nu + 1) * 4));
  HIP_TRY(ctx->b_vals1.ensure(n * 4));
  HIP_TRY(hipMemcpyAsync(ctx->b_vals0.p, row_start.data(), (nu + 1) * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(ctx->b_vals1.p, order.data(), n * 4, hipMemcpyHostToDevice, s));
  rc = insert_and_lookup(ctx, nu);
  if (rc) return rc;
  hipLaunchKernelGGL(k_merge_sums, dim3((unsigned)nu), dim3(256), 0, s, m, ctx->b_rank.as<uint32_t>(),
                     ctx->b_vals0.as<uint32_t>(), ctx->b_vals1.as<uint32_t>(), d_sums, apply_caps, truncation_distance,
                     max_weight, ctx->d_state);
  rc = sync_state(ctx);  // also keeps the host vectors alive until the copies are done
  if (rc) return rc;
  return check_state_error(ctx);
}


// ---- Layer serialization -----------------------------------------------------------------
int vbx_blocks_serialize(vbx_ctx* ctx, int layer, const int32_t* idx, size_t n, uint32_t* words, uint8_t* has_data) {
  if (!ctx || (n && (!idx || !words))) return VBX_ERR_INVALID;
  if (layer != VBX_LAYER_TSDF && layer != VBX_LAYER_ESDF) return VBX_ERR_INVALID;
  HIP_TRY(hipSetDevice(ctx->device));
  if (n == 0) return VBX_OK;
  if (layer == VBX_LAYER_ESDF && !ctx->esdf_init) {
    ctx->fail("ESDF layer is empty");
    return VBX_ERR_INVALID;
  }
  int rc = upload_idx(ctx, idx, n);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  MapDev& m = ctx->map;
  const size_t wpb = (size_t)m.nvox * (layer == VBX_LAYER_TSDF ? 3 : 2);
  HIP_TRY(ctx->b_keys0.ensure(n * wpb * 4));
  HIP_TRY(ctx->b_vals0.ensure(n * 4));
  hipLaunchKernelGGL(k_lookup_slots_flags, grid_for(n), dim3(256), 0, s, m, ctx->b_head.as<int32_t>(), (uint32_t)n,
                     layer == VBX_LAYER_TSDF ? kFlagPublished : kFlagEsdfAlloc, ctx->b_rank.as<uint32_t>(),
                     ctx->b_vals0.as<uint32_t>());
  if (layer == VBX_LAYER_TSDF)
    hipLaunchKernelGGL(k_serialize_tsdf, dim3((unsigned)n), dim3(256), 0, s, m, ctx->b_rank.as<uint32_t>(),
                       ctx->b_keys0.as<uint32_t>());
  else
    hipLaunchKernelGGL(k_serialize_esdf, dim3((unsigned)n), dim3(256), 0, s, m.nvox, ctx->b_edist.as<float>(),
                       ctx->b_estate.as<uint32_t>(), ctx->b_rank.as<uint32_t>(), ctx->b_keys0.as<uint32_t>());
  std::vector<uint32_t> flags(n);
  HIP_TRY(hipMemcpyAsync(flags.data(), ctx->b_vals0.p, n * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(words, ctx->b_keys0.p, n * wpb * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (size_t i = 0; i < n; ++i) {
    if (flags[i] == ~0u) {
      ctx->fail("block (%d,%d,%d) is not allocated", idx[3 * i], idx[3 * i + 1], idx[3 * i + 2]);
      return VBX_ERR_INVALID;
    }
    if (has_data) has_data[i] = (layer == VBX_LAYER_TSDF && (flags[i] & kFlagHasData)) ? 1 : 0;
  }
  return VBX_OK;
}

int vbx_blocks_deserialize(vbx_ctx* ctx, int layer, const int32_t* idx, size_t n, const uint32_t* words,
                           const uint8_t* has_data) {
  if (!ctx || (n && (!idx || !words))) return VBX_ERR_INVALID;
  if (layer != VBX_LAYER_TSDF && layer != VBX_LAYER_ESDF) return VBX_ERR_INVALID;
  HIP_TRY(hipSetDevice(ctx->device));
  if (n == 0) return VBX_OK;
  int rc = upload_idx(ctx, idx, n);
  if (rc) return rc;
  if (layer == VBX_LAYER_ESDF) {
    rc = esdf_ensure(ctx);
    if (rc) return rc;
  }
  hipStream_t s = ctx->stream;
  MapDev& m = ctx->map;
  const size_t wpb = (size_t)m.nvox * (layer == VBX_LAYER_TSDF ? 3 : 2);
  HIP_TRY(ctx->b_keys0.ensure(n * wpb * 4));
  HIP_TRY(hipMemcpyAsync(ctx->b_keys0.p, words, n * wpb * 4, hipMemcpyHostToDevice, s));
  uint8_t* d_hd = nullptr;
  if (has_data) {
    HIP_TRY(ctx->b_graze.ensure(n));
    HIP_TRY(hipMemcpyAsync(ctx->b_graze.p, has_data, n, hipMemcpyHostToDevice, s));
    d_hd = ctx->b_graze.as<uint8_t>();
  }
  rc = insert_and_lookup(ctx, n);
  if (rc) return rc;
  if (layer == VBX_LAYER_TSDF) {
    hipLaunchKernelGGL(k_deserialize_tsdf, dim3((unsigned)n), dim3(256), 0, s, m, ctx->b_rank.as<uint32_t>(),
                       ctx->b_keys0.as<uint32_t>());
    hipLaunchKernelGGL(k_set_block_flags, grid_for(n), dim3(256), 0, s, m, ctx->b_rank.as<uint32_t>(), (uint32_t)n,
                       kFlagPublished | kFlagUpdMask, d_hd, kFlagHasData);
  } else {
    hipLaunchKernelGGL(k_deserialize_esdf, dim3((unsigned)n), dim3(256), 0, s, m.nvox, ctx->b_edist.as<float>(),
                       ctx->b_estate.as<uint32_t>(), ctx->b_rank.as<uint32_t>(), ctx->b_keys0.as<uint32_t>());
    hipLaunchKernelGGL(k_set_block_flags, grid_for(n), dim3(256), 0, s, m, ctx->b_rank.as<uint32_t>(), (uint32_t)n,
                       // voxels written from outside: not a fixed point of this map's wavefront (the next order-free update relaxes the
                       // whole block, not its shell) and news for the host mirror (VBX_UPDATE_DIRTY)
                       kFlagEsdfAlloc | (kFlagUpdMask << kFlagEsdfUpdShift) | kFlagEsdfUnsettled | kFlagEsdfDirty, (const uint8_t*)nullptr, 0u);
  }
  rc = sync_state(ctx);
  if (rc) return rc;
  return check_state_error(ctx);
}

// FastTsdfIntegrator's process-wide reset counter (tsdf_integrator.cc:564)
int64_t vbx_fast_reset_counter_get(void) { return g_fast_reset_counter.load(); }
void vbx_fast_reset_counter_set(int64_t value) { g_fast_reset_counter.store(value); }

// Self-test hook of the hand-written stable radix sort (vbx_sort.hpp): n pseudo-random keys,
// sorted on bits [begin_bit, end_bit) on the device, compared with std::stable_sort on the host.
int vbx_selftest_sort(vbx_ctx* ctx, uint32_t n, uint32_t begin_bit, uint32_t end_bit, uint32_t seed, int with_vals) {
  if (!ctx || begin_bit > end_bit || end_bit > 64) return VBX_ERR_INVALID;
  HIP_TRY(hipSetDevice(ctx->device));
  std::vector<uint64_t> keys(n), ref(n), got(n);
  std::vector<uint32_t> vals(n), gotv(n), idx(n);
  uint64_t x = 0x9E3779B97F4A7C15ull ^ ((uint64_t)seed << 17) ^ n;
  const bool narrow = (seed & 1u) != 0;  // odd seeds: few distinct field values -> long equal runs
  for (uint32_t i = 0; i < n; ++i) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    keys[i] = narrow ? (x & ~(0xFFFFFFF0ull << (begin_bit < 60 ? begin_bit : 60))) : x;
    vals[i] = i;
    idx[i] = i;
  }
  const unsigned bits = end_bit - begin_bit;
  const uint64_t fmask = bits >= 64 ? ~0ull : ((1ull << bits) - 1);
  auto field = [&](uint64_t k) { return bits ? ((k >> begin_bit) & fmask) : 0ull; };
  std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return field(keys[a]) < field(keys[b]); });
  HIP_TRY(ctx->b_keys0.ensure(std::max<size_t>(n, 1) * 8));
  HIP_TRY(ctx->b_vals0.ensure(std::max<size_t>(n, 1) * 4));
  HIP_TRY(ctx->b_keys1.ensure(std::max<size_t>(n, 1) * 8));
  HIP_TRY(ctx->b_vals1.ensure(std::max<size_t>(n, 1) * 4));
  if (n) {
    HIP_TRY(hipMemcpyAsync(ctx->b_keys0.p, keys.data(), (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->b_vals0.p, vals.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  int rc = stable_sort01(ctx, n, begin_bit, end_bit, with_vals != 0);
  if (rc) return rc;
  if (n) {
    HIP_TRY(hipMemcpyAsync(got.data(), ctx->b_keys1.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (with_vals) HIP_TRY(hipMemcpyAsync(gotv.data(), ctx->b_vals1.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (uint32_t i = 0; i < n; ++i) {
    if (got[i] != keys[idx[i]] || (with_vals && gotv[i] != idx[i])) {
      ctx->fail("stable sort self-test: position %u holds key %llx (value %u), expected key %llx (value %u)", i,
                (unsigned long long)got[i], with_vals ? gotv[i] : 0u, (unsigned long long)keys[idx[i]], idx[i]);
      return VBX_ERR_HIP;
    }
  }
  return VBX_OK;
}

int vbx_selftest_scan(vbx_ctx* ctx, uint32_t n, uint32_t seed, uint32_t repeats) {
  if (!ctx) return VBX_ERR_INVALID;
  HIP_TRY(hipSetDevice(ctx->device));
  std::vector<uint32_t> in(n), got(n);
  uint64_t x = 0x9E3779B97F4A7C15ull ^ ((uint64_t)seed << 21) ^ n;
  for (uint32_t i = 0; i < n; ++i) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    in[i] = (seed & 1u) ? (uint32_t)(x & 1u) : (uint32_t)(x % 1000u);
  }
  HIP_TRY(ctx->b_vals0.ensure(std::max<size_t>(n, 1) * 4));
  HIP_TRY(ctx->b_vals1.ensure(std::max<size_t>(n, 1) * 4));
  if (n) HIP_TRY(hipMemcpyAsync(ctx->b_vals0.p, in.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  for (uint32_t rep = 0; rep < std::max<uint32_t>(repeats, 1); ++rep) {  // back-to-back calls reuse the descriptors
    int rc = exclusive_scan_u32(ctx, ctx->b_vals0.as<uint32_t>(), ctx->b_vals1.as<uint32_t>(), n);
    if (rc) return rc;
  }
  if (n) HIP_TRY(hipMemcpyAsync(got.data(), ctx->b_vals1.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  uint32_t run = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (got[i] != run) {
      ctx->fail("exclusive scan self-test: position %u holds %u, expected %u (n = %u)", i, got[i], run, n);
      return VBX_ERR_HIP;
    }
    run += in[i];
  }
  return VBX_OK;
}

void vbx_mesh_cfg_default(vbx_mesh_cfg* cfg) {
  if (!cfg) return;
  cfg->use_color = 1;        // mesh_integrator.h:50
  cfg->min_weight = 1e-4f;   // mesh_integrator.h:51
}

int vbx_mesh_generate(vbx_ctx* ctx, const vbx_mesh_cfg* cfg, int only_mesh_updated_blocks, int clear_updated_flag,
                      size_t* n_blocks, size_t* n_vertices) {
  if (!ctx || !cfg) return VBX_ERR_INVALID;
  int rc = mesh_generate(ctx, cfg, only_mesh_updated_blocks, clear_updated_flag);
  prof_collect(ctx);
  if (rc) return rc;
  if (n_blocks) *n_blocks = ctx->mesh_off.size() - 1;
  if (n_vertices) *n_vertices = (size_t)ctx->mesh_off.back() * 3;
  return VBX_OK;
}

int vbx_mesh_blocks(vbx_ctx* ctx, int32_t* idx_xyz, uint64_t* vertex_offset, size_t cap, size_t* n) {
  if (!ctx || !n) return VBX_ERR_INVALID;
  const size_t nb = ctx->mesh_off.size() - 1;
  *n = nb;
  if (cap < nb) return idx_xyz || vertex_offset ? VBX_ERR_CAPACITY : VBX_OK;
  if (idx_xyz && nb) std::memcpy(idx_xyz, ctx->mesh_idx.data(), nb * 12);
  if (vertex_offset)
    for (size_t i = 0; i <= nb; ++i) vertex_offset[i] = (uint64_t)ctx->mesh_off[i] * 3;
  return VBX_OK;
}

int vbx_mesh_download(vbx_ctx* ctx, float* vertices, float* normals, uint8_t* rgba, size_t cap_vertices) {
  if (!ctx) return VBX_ERR_INVALID;
  const size_t nv = (size_t)ctx->mesh_off.back() * 3;
  if (cap_vertices < nv) {
    ctx->fail("vbx_mesh_download: room for %zu vertices, the mesh has %zu", cap_vertices, nv);
    return VBX_ERR_CAPACITY;
  }
  if (rgba && !ctx->mesh_has_colors) {
    ctx->fail("vbx_mesh_download: the last vbx_mesh_generate ran with use_color = 0");
    return VBX_ERR_INVALID;
  }
  if (nv == 0) return VBX_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  if (vertices) HIP_TRY(hipMemcpyAsync(vertices, ctx->b_mesh_verts.p, nv * 12, hipMemcpyDeviceToHost, ctx->stream));
  if (normals) HIP_TRY(hipMemcpyAsync(normals, ctx->b_mesh_normals.p, nv * 12, hipMemcpyDeviceToHost, ctx->stream));
  if (rgba) HIP_TRY(hipMemcpyAsync(rgba, ctx->b_mesh_colors.p, nv * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return VBX_OK;
}

int vbx_mesh_device_ptrs(vbx_ctx* ctx, const float** d_vertices, const float** d_normals, const uint8_t** d_rgba) {
  if (!ctx) return VBX_ERR_INVALID;
  if (d_vertices) *d_vertices = ctx->b_mesh_verts.as<float>();
  if (d_normals) *d_normals = ctx->b_mesh_normals.as<float>();
  if (d_rgba) *d_rgba = ctx->mesh_has_colors ? ctx->b_mesh_colors.as<uint8_t>() : nullptr;
  return VBX_OK;
}

int vbx_selftest_unordered_order(uint32_t n, uint32_t seed, uint32_t hash_mask) {
  // pseudo-random 32-bit hashes (masked to force bucket collisions when asked)
  std::vector<uint32_t> h(n);
  uint64_t x = 0x9E3779B97F4A7C15ull ^ ((uint64_t)seed << 17);
  for (uint32_t i = 0; i < n; ++i) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    h[i] = (uint32_t)(x >> 20) & hash_mask;
  }
  std::vector<uint32_t> fast, real, seq, runs;
  std::vector<int32_t> head, nxt;
  const auto t0 = std::chrono::steady_clock::now();
  unordered_iteration_order(h.data(), n, &fast, OrderScratch{&seq, &runs, &head, &nxt});
  const auto t1 = std::chrono::steady_clock::now();
  unordered_iteration_order(h.data(), n, &fast, OrderScratch{&seq, &runs, &head, &nxt});
  const auto t2 = std::chrono::steady_clock::now();
  unordered_iteration_order_real(h.data(), n, &real);
  const auto t3 = std::chrono::steady_clock::now();
  if (getenv("VBX_DEBUG_MERGED"))
    fprintf(stderr, "order of %u keys: arrays %.1f us (first call %.1f), container %.1f us\n", n,
            std::chrono::duration<double, std::micro>(t2 - t1).count(),
            std::chrono::duration<double, std::micro>(t1 - t0).count(),
            std::chrono::duration<double, std::micro>(t3 - t2).count());
  if (fast.size() != real.size()) return VBX_ERR_HIP;
  for (size_t i = 0; i < fast.size(); ++i)
    if (fast[i] != real[i]) return VBX_ERR_HIP;
  return VBX_OK;
}

int vbx_get_counters(vbx_ctx* ctx, vbx_counters* out) {
  if (!ctx || !out) return VBX_ERR_INVALID;
  *out = ctx->counters;
  return VBX_OK;
}
int vbx_enable_timing(vbx_ctx* ctx, int enable) {
  if (!ctx) return VBX_ERR_INVALID;
  ctx->timing = enable != 0;
  return VBX_OK;
}
int vbx_get_timing(vbx_ctx* ctx, vbx_timing* out) {
  if (!ctx || !out) return VBX_ERR_INVALID;
  *out = ctx->last_timing;
  return VBX_OK;
}
int vbx_set_pool_limit(vbx_ctx* ctx, uint32_t max_blocks_limit) {
  if (!ctx) return VBX_ERR_INVALID;
  ctx->pool_limit = max_blocks_limit;
  return VBX_OK;
}
int vbx_profile_enable(vbx_ctx* ctx, int enable) {
  if (!ctx) return VBX_ERR_INVALID;
  prof_collect(ctx);
  ctx->prof = enable != 0;
  return VBX_OK;
}
int vbx_profile_reset(vbx_ctx* ctx) {
  if (!ctx) return VBX_ERR_INVALID;
  prof_collect(ctx);
  ctx->prof_table.clear();
  ctx->prof_calls = 0;
  return VBX_OK;
}
int vbx_profile_get(vbx_ctx* ctx, char* buf, size_t cap, size_t* needed, uint64_t* calls) {
  if (!ctx) return VBX_ERR_INVALID;
  prof_collect(ctx);
  std::string out;
  char line[256];
  for (const auto& kv : ctx->prof_table) {
    snprintf(line, sizeof(line), "%s\t%llu\t%.6f\n", kv.first.c_str(), (unsigned long long)kv.second.first,
             kv.second.second);
    out += line;
  }
  if (needed) *needed = out.size() + 1;
  if (calls) *calls = ctx->prof_calls;
  if (buf && cap) {
    const size_t n = std::min(out.size(), cap - 1);
    std::memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return VBX_OK;
}

}  // extern "C"

#ifdef VBX_ESDF_STATS
// measurement build only (tools/esdf_tile_stats.py)
extern "C" int vbx_debug_esdf_stats(unsigned long long out[16], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_esdf_stats), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_esdf_stats), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

#ifdef VBX_SORT_STATS
// measurement build only (tools/sort_stats.py)
extern "C" int vbx_debug_sort_stats(unsigned long long out[16], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sort_stats), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_sort_stats), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
