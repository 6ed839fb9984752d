// mesh_integrator_hip.cc — the body behind mesh_integrator_hip.h's specialisation of
// MeshIntegrator<TsdfVoxel>::generateMesh (include/voxblox/mesh/mesh_integrator.h:142-172).
#include "mesh_integrator_hip.h"

#include <cstring>
#include <numeric>

#include "device_mirror.h"
#include "voxblox/utils/timing.h"

namespace voxblox {
namespace hip {

void generateMeshOnDevice(const MeshIntegratorConfig& config, const Layer<TsdfVoxel>* sdf_layer_const,
                          Layer<TsdfVoxel>* sdf_layer_mutable, MeshLayer* mesh_layer, bool only_mesh_updated_blocks,
                          bool clear_updated_flag) {
  static_assert(sizeof(Point) == 12 && sizeof(Color) == 4, "Mesh::vertices / normals / colors are filled with memcpy");
  timing::Timer mesh_timer("mesh/generate");
  // the mirror is keyed by the layer's address; reconcile only reads the layer (a const MeshIntegrator may hand in a const one)
  Layer<TsdfVoxel>* layer = sdf_layer_mutable ? sdf_layer_mutable : const_cast<Layer<TsdfVoxel>*>(sdf_layer_const);
  MirrorRef pinned = mirrorOf(layer);
  DeviceMirror& dev = *pinned;
  // The block list is the HOST layer's (:147-152): the host is the source of truth between calls.
  BlockIndexList all_tsdf_blocks;
  if (only_mesh_updated_blocks) sdf_layer_const->getAllUpdatedBlocks(Update::kMesh, &all_tsdf_blocks);
  else sdf_layer_const->getAllAllocatedBlocks(&all_tsdf_blocks);
  // Allocate all the mesh memory (:154-157) — in the reference's sequence, so the MeshLayer's container iterates alike.
  for (const BlockIndex& block_index : all_tsdf_blocks) mesh_layer->allocateMeshPtrByIndex(block_index);
  // Host edits since the last drop-in call (loaded blocks, removed blocks, Update bits set by mergeBlock) reach the device
  // first.  A kMesh bit the HOST cleared (somebody meshed with another integrator) does not travel through reconcile —
  // consumers clearing bits is the normal case and costs nothing there — so the two lists are compared here and the
  // blocks the device would mesh in excess go up again with the host's bits (rare).
  reconcileTsdfFromHost(dev, layer);
  if (only_mesh_updated_blocks) {
    size_t n_dev = 0;
    CHECK_EQ(vbx_blocks_updated(dev.ctx, VBX_LAYER_TSDF, VBX_UPDATE_MESH, nullptr, 0, &n_dev), VBX_OK) << vbx_last_error(dev.ctx);
    std::vector<int32_t> dev_idx(3 * n_dev);
    if (n_dev)
      CHECK_EQ(vbx_blocks_updated(dev.ctx, VBX_LAYER_TSDF, VBX_UPDATE_MESH, dev_idx.data(), n_dev, &n_dev), VBX_OK)
          << vbx_last_error(dev.ctx);
    AnyIndexHashMapType<int>::type host_set;
    for (const BlockIndex& bi : all_tsdf_blocks) host_set[bi] = 1;
    bool again = false;
    for (size_t i = 0; i < n_dev; ++i) {
      const BlockIndex bi(dev_idx[3 * i], dev_idx[3 * i + 1], dev_idx[3 * i + 2]);
      if (host_set.count(bi)) continue;
      markBlockEdited(layer, bi);   // goes up with the host's bits (kMesh cleared)
      again = true;
    }
    if (again) reconcileTsdfFromHost(dev, layer);
  }
  vbx_mesh_cfg cfg;
  vbx_mesh_cfg_default(&cfg);
  cfg.use_color = config.use_color ? 1 : 0;
  cfg.min_weight = config.min_weight;
  size_t n_blocks = 0, n_vertices = 0;
  CHECK_EQ(vbx_mesh_generate(dev.ctx, &cfg, only_mesh_updated_blocks ? 1 : 0, clear_updated_flag ? 1 : 0, &n_blocks, &n_vertices), VBX_OK)
      << vbx_last_error(dev.ctx);
  std::vector<int32_t> idx(3 * n_blocks + 3);
  std::vector<uint64_t> off(n_blocks + 1, 0);
  size_t n = 0;
  CHECK_EQ(vbx_mesh_blocks(dev.ctx, idx.data(), off.data(), n_blocks, &n), VBX_OK) << vbx_last_error(dev.ctx);
  // vertices, normals (12 B stride) and colours of all meshed blocks in one copy each
  float* vtx = static_cast<float*>(dev.down_staging.ensure(n_vertices * (12 + 12 + 4) + 64));
  float* nrm = vtx + 3 * n_vertices;
  uint8_t* rgba = reinterpret_cast<uint8_t*>(nrm + 3 * n_vertices);
  if (n_vertices)
    CHECK_EQ(vbx_mesh_download(dev.ctx, vtx, nrm, config.use_color ? rgba : nullptr, n_vertices), VBX_OK) << vbx_last_error(dev.ctx);
  AnyIndexHashMapType<size_t>::type meshed;   // block -> position in the device's table
  for (size_t b = 0; b < n; ++b) meshed[BlockIndex(idx[3 * b], idx[3 * b + 1], idx[3 * b + 2])] = b;
  for (const BlockIndex& block_index : all_tsdf_blocks) {
    // updateMeshForBlock (:250-270): the Mesh is cleared, refilled, marked updated — also when it ends up empty
    Mesh::Ptr mesh = mesh_layer->getMeshPtrByIndex(block_index);
    mesh->clear();
    auto it = meshed.find(block_index);
    // (a listed block the device does not hold cannot happen after reconcile; the reference logs an error and goes on, :257-261)
    if (it == meshed.end()) {
      LOG(ERROR) << "Trying to mesh a non-existent block at index: " << block_index.transpose();
      continue;
    }
    const size_t b = it->second;
    const size_t first = off[b], count = off[b + 1] - off[b];
    mesh->vertices.resize(count);
    mesh->normals.resize(count);
    if (count) {
      std::memcpy(static_cast<void*>(mesh->vertices[0].data()), vtx + 3 * first, count * 12);
      std::memcpy(static_cast<void*>(mesh->normals[0].data()), nrm + 3 * first, count * 12);
    }
    mesh->indices.resize(count);                                         // marching_cubes.h:94-96: 0, 1, 2, ... per block
    std::iota(mesh->indices.begin(), mesh->indices.end(), 0);
    if (config.use_color) {                                              // updateMeshColor (:372-392)
      mesh->colors.resize(count);
      if (count) std::memcpy(static_cast<void*>(&mesh->colors[0]), rgba + 4 * first, count * 4);
    }
    mesh->updated = true;                                                // :269
    if (clear_updated_flag) {                                            // generateMeshBlocksFunction :176-183
      Block<TsdfVoxel>::Ptr block = sdf_layer_mutable->getBlockPtrByIndex(block_index);
      block->updated().reset(Update::kMesh);
      HostBlockRecord<TsdfVoxel>* rec = dev.tsdf_known.find(block_index);
      if (rec) rec->bits = static_cast<uint8_t>(block->updated().to_ulong());
    }
  }
  mesh_timer.Stop();
}

}  // namespace hip
}  // namespace voxblox
