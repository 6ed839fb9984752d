// vbx_host_mesh.hpp — host driver of the incremental mesher (see vbx_kernels_mesh.hpp)
// Part of libvbx_hip.so's single translation unit (included by vbx_hip.hip, in order).
namespace {

template <int VPS>
int mesh_generate_t(vbx_ctx* ctx, const vbx_mesh_cfg* cfg, int only_updated, int clear_flag) {
  hipStream_t s = ctx->stream;
  MapDev& m = ctx->map;
  int rc = sync_state(ctx);
  if (rc) return rc;
  const uint32_t used = ctx->h_state.pool_used;
  ctx->mesh_idx.clear();
  ctx->mesh_off.assign(1, 0u);
  ctx->mesh_has_colors = cfg->use_color != 0;
  if (used == 0) return VBX_OK;
  const size_t n1 = (size_t)used + 1;
  HIP_TRY(ctx->b_head.ensure(n1 * 4)); HIP_TRY(ctx->b_rank.ensure(n1 * 4));
  HIP_TRY(ctx->b_mesh_list.ensure(n1 * 4)); HIP_TRY(ctx->b_mesh_cnt.ensure(n1 * 4));
  HIP_TRY(ctx->b_mesh_off.ensure(n1 * 4));
  HIP_TRY(ctx->b_mesh_tab.ensure(n1 * 16));
  uint32_t* head = ctx->b_head.as<uint32_t>();
  uint32_t* rank = ctx->b_rank.as<uint32_t>();
  KLAUNCH(k_mesh_select, grid_for(n1), dim3(256), 0, s, m, used, only_updated, head);
  rc = exclusive_scan_u32(ctx, head, rank, n1);
  if (rc) return rc;
  KLAUNCH(k_mesh_compact, grid_for(n1), dim3(256), 0, s, used, head, rank, ctx->b_mesh_list.as<uint32_t>(),
                     ctx->b_mesh_cnt.as<uint32_t>());
  MeshDev d{};
  d.list = ctx->b_mesh_list.as<uint32_t>();
  d.n_list = rank + used;
  d.tri_count = ctx->b_mesh_cnt.as<uint32_t>();
  d.tri_off = ctx->b_mesh_off.as<uint32_t>();
  d.min_weight = cfg->min_weight;
  d.block_size = m.voxel_size * (float)m.vps;                 // layer.h:39
  d.block_size_inv = (float)(1.0 / (double)d.block_size);     // layer.h:41
  // pass 1 over every pool slot's worth of workgroups: the number of selected blocks is still on
  // the device, surplus workgroups leave at once
  KLAUNCH((k_mesh_block<VPS, false>), dim3(used), dim3(MeshThreads<VPS>::value), 0, s, m, d);
  rc = exclusive_scan_u32(ctx, ctx->b_mesh_cnt.as<uint32_t>(), ctx->b_mesh_off.as<uint32_t>(), n1);
  if (rc) return rc;
  const uint32_t* const ex[3] = {rank + used, ctx->b_mesh_off.as<uint32_t>() + used, nullptr};
  uint32_t got[3] = {0, 0, 0};
  rc = sync_state3(ctx, ex, got);
  if (rc) return rc;
  const uint32_t n_list = got[0], n_tri = got[1];
  if (n_list == 0) return VBX_OK;
  HIP_TRY(ctx->b_mesh_verts.ensure(std::max<size_t>(n_tri, 1) * 36));
  HIP_TRY(ctx->b_mesh_normals.ensure(std::max<size_t>(n_tri, 1) * 36));
  if (cfg->use_color) HIP_TRY(ctx->b_mesh_colors.ensure(std::max<size_t>(n_tri, 1) * 12));
  d.verts = ctx->b_mesh_verts.as<float>();
  d.normals = ctx->b_mesh_normals.as<float>();
  d.colors = cfg->use_color ? ctx->b_mesh_colors.as<uint32_t>() : nullptr;
  if (n_tri) KLAUNCH((k_mesh_block<VPS, true>), dim3(n_list), dim3(MeshThreads<VPS>::value), 0, s, m, d);
  int32_t* tab_idx = ctx->b_mesh_tab.as<int32_t>();
  uint32_t* tab_off = reinterpret_cast<uint32_t*>(tab_idx + 3 * n1);
  KLAUNCH(k_mesh_finish, grid_for((size_t)n_list + 1), dim3(256), 0, s, m, d, clear_flag, tab_idx, tab_off);
  ctx->mesh_idx.resize((size_t)n_list * 3);
  ctx->mesh_off.resize((size_t)n_list + 1);
  HIP_TRY(hipMemcpyAsync(ctx->mesh_idx.data(), tab_idx, (size_t)n_list * 12, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(ctx->mesh_off.data(), tab_off, ((size_t)n_list + 1) * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  return VBX_OK;
}

int mesh_generate(vbx_ctx* ctx, const vbx_mesh_cfg* cfg, int only_updated, int clear_flag) {
  HIP_TRY(hipSetDevice(ctx->device));
  switch (ctx->map.vps) {
    case 8: return mesh_generate_t<8>(ctx, cfg, only_updated, clear_flag);
    case 16: return mesh_generate_t<16>(ctx, cfg, only_updated, clear_flag);
    default:
      ctx->fail("mesh: voxels_per_side must be 8 or 16 (LDS tile)");
      return VBX_ERR_UNSUPPORTED;
  }
}

}  // namespace
