// Exercises the C++ host shim (voxblox_amd/host/vbx_integrators.hpp) the way
// voxblox's test_sdf_integrators.cc drives the reference classes: build the three TSDF
// integrators through the factory, integrate a small synthetic wall, run the ESDF
// integrator incrementally, and print a checksum line that the pytest driver compares with
// the oracle.  Needs a GPU (no CPU fallback).
#include <cmath>
#include <cstdio>

#include "../../voxblox_amd/host/vbx_integrators.hpp"
#include "../../voxblox_amd/host/vbx_io.hpp"
#include "../../voxblox_amd/host/vbx_mesh.hpp"
#include "../../voxblox_amd/host/vbx_sharded.hpp"

using namespace vbx_host;

int main() {
  const float voxel = 0.1f;
  auto map = std::make_shared<DeviceMap>(voxel, 16, 2048, 0);
  Layer<TsdfVoxel> tsdf(map);
  Layer<EsdfVoxel> esdf(map);

  TsdfIntegratorBase::Config config;
  config.default_truncation_distance = 4 * voxel;
  config.integrator_threads = 1;

  // a wall at z = 3 m seen through a 64x48 pinhole, f = 32
  Pointcloud points;
  Colors colors;
  for (int v = 0; v < 48; ++v)
    for (int u = 0; u < 64; ++u) {
      const double x = (u + 0.5 - 32) / 32.0, y = (v + 0.5 - 24) / 32.0;
      points.push_back({static_cast<float>(3.0 * x), static_cast<float>(3.0 * y), 3.0f});
      Color c;
      c.r = static_cast<uint8_t>(u); c.g = static_cast<uint8_t>(v); c.b = 40; c.a = 255;
      colors.push_back(c);
    }
  Transformation T_G_C;

  for (const std::string& name : kTsdfIntegratorTypeNames) {
    tsdf.removeAllBlocks();
    TsdfIntegratorBase::Ptr integrator = TsdfIntegratorFactory::create(name, config, &tsdf);
    integrator->integratePointCloud(T_G_C, points, colors);
    BlockIndexList blocks;
    tsdf.getAllAllocatedBlocks(&blocks);
    size_t observed = 0;
    double sum_w = 0, sum_d = 0;
    for (const BlockIndex& b : blocks) {
      auto blk = tsdf.getBlockPtrByIndex(b);
      if (!blk) return 2;
      if (!blk->updated(Update::kMap) || !blk->updated(Update::kMesh) || !blk->updated(Update::kEsdf)) return 3;
      for (size_t i = 0; i < blk->num_voxels(); ++i) {
        const TsdfVoxel& vx = blk->getVoxelByLinearIndex(i);
        if (vx.weight > 1e-6f) {
          ++observed;
          sum_w += vx.weight;
          sum_d += vx.distance;
        }
      }
    }
    std::printf("%s blocks=%zu observed=%zu sum_w=%.6f sum_d=%.6f\n", name.c_str(), blocks.size(), observed, sum_w,
                sum_d);
    if (blocks.empty() || observed == 0) return 4;
  }

  // MeshIntegrator the way tsdf_server.cc:104-106 / :446-449 drives it: incremental, clearing kMesh
  {
    MeshLayer mesh_layer(tsdf.block_size());
    MeshIntegratorConfig mesh_config;
    MeshIntegrator<TsdfVoxel> mesh_integrator(mesh_config, &tsdf, &mesh_layer);
    BlockIndexList flagged;
    tsdf.getAllUpdatedBlocks(Update::kMesh, &flagged);
    mesh_integrator.generateMesh(true, true);
    BlockIndexList meshes, updated_meshes, still_flagged_mesh;
    mesh_layer.getAllAllocatedMeshes(&meshes);
    mesh_layer.getAllUpdatedMeshes(&updated_meshes);
    tsdf.getAllUpdatedBlocks(Update::kMesh, &still_flagged_mesh);
    size_t n_vert = 0;
    double sum_z = 0;
    for (const BlockIndex& b : meshes) {
      const Mesh& mesh = mesh_layer.getMeshByIndex(b);
      if (mesh.vertices.size() != mesh.normals.size() || mesh.vertices.size() != mesh.colors.size() ||
          mesh.vertices.size() != mesh.indices.size() || mesh.vertices.size() % 3)
        return 20;
      n_vert += mesh.vertices.size();
      for (const Point& v : mesh.vertices) sum_z += v.z;
    }
    std::printf("mesh blocks=%zu vertices=%zu mean_z=%.6f\n", meshes.size(), n_vert, n_vert ? sum_z / n_vert : 0.0);
    // the wall is at z = 3: every vertex of the zero crossing lies within a voxel of it
    if (meshes.size() != flagged.size() || updated_meshes.size() != meshes.size() || !still_flagged_mesh.empty() ||
        n_vert == 0 || std::fabs(sum_z / n_vert - 3.0) > voxel)
      return 21;
    mesh_integrator.generateMesh(true, true);  // nothing flagged any more: the layer is unchanged
    BlockIndexList again;
    mesh_layer.getAllAllocatedMeshes(&again);
    if (again.size() != meshes.size()) return 22;
  }

  EsdfIntegrator::Config esdf_config;
  esdf_config.min_distance_m = config.default_truncation_distance / 2;
  EsdfIntegrator esdf_integrator(esdf_config, &tsdf, &esdf);
  esdf_integrator.updateFromTsdfLayer(true);
  BlockIndexList eblocks, still_flagged;
  esdf.getAllAllocatedBlocks(&eblocks);
  tsdf.getAllUpdatedBlocks(Update::kEsdf, &still_flagged);
  size_t eobs = 0, efixed = 0;
  for (const BlockIndex& b : eblocks) {
    auto blk = esdf.getBlockPtrByIndex(b);
    if (!blk) return 5;
    for (size_t i = 0; i < blk->num_voxels(); ++i) {
      eobs += blk->getVoxelByLinearIndex(i).observed;
      efixed += blk->getVoxelByLinearIndex(i).fixed;
    }
  }
  std::printf("esdf blocks=%zu observed=%zu fixed=%zu tsdf_blocks_still_flagged=%zu\n", eblocks.size(), eobs, efixed,
              still_flagged.size());
  if (eblocks.size() != tsdf.getNumberOfAllocatedBlocks() || eobs == 0 || efixed == 0 || !still_flagged.empty())
    return 6;
  // save_map / load_map path: TSDF + ESDF sections in one file, reloaded into a fresh map
  const char* path = "/tmp/vbx_shim_demo.voxblox";
  if (!io::SaveLayer(tsdf, path, true) || !io::SaveLayer(esdf, path, false)) return 7;
  auto map2 = std::make_shared<DeviceMap>(voxel, 16, 2048, 0);
  Layer<TsdfVoxel> tsdf2(map2);
  Layer<EsdfVoxel> esdf2(map2);
  if (!io::LoadBlocksFromFile(path, false, &tsdf2) || !io::LoadBlocksFromFile(path, true, &esdf2)) return 8;
  BlockIndexList b1, b2;
  tsdf.getAllAllocatedBlocks(&b1);
  tsdf2.getAllAllocatedBlocks(&b2);
  if (b1.size() != b2.size() || esdf2.getNumberOfAllocatedBlocks() != eblocks.size()) return 9;
  size_t diff = 0;
  for (size_t i = 0; i < b1.size(); ++i) {
    auto x = tsdf.getBlockPtrByIndex(b1[i]);
    auto y = tsdf2.getBlockPtrByIndex(b1[i]);
    if (!x || !y) return 10;
    diff += std::memcmp(&x->getVoxelByLinearIndex(0), &y->getVoxelByLinearIndex(0), x->num_voxels() * sizeof(TsdfVoxel)) != 0;
    auto ex = esdf.getBlockPtrByIndex(b1[i]);
    auto ey = esdf2.getBlockPtrByIndex(b1[i]);
    if (!ex || !ey) return 11;
    for (size_t v = 0; v < ex->num_voxels(); ++v) {
      const EsdfVoxel& p = ex->getVoxelByLinearIndex(v);
      const EsdfVoxel& q = ey->getVoxelByLinearIndex(v);
      diff += std::memcmp(&p.distance, &q.distance, 4) != 0 || p.observed != q.observed || p.fixed != q.fixed;
    }
  }
  std::printf("io round trip blocks=%zu differing=%zu\n", b1.size(), diff);
  if (diff) return 12;
  // bulk mirror == per-block mirror, both layers
  std::vector<std::shared_ptr<Block<TsdfVoxel>>> bulk_t;
  std::vector<std::shared_ptr<Block<EsdfVoxel>>> bulk_e;
  tsdf.getBlocksByIndex(b1, &bulk_t);
  esdf.getBlocksByIndex(b1, &bulk_e);
  if (bulk_t.size() != b1.size() || bulk_e.size() != b1.size()) return 13;
  for (size_t i = 0; i < b1.size(); ++i) {
    auto x = tsdf.getBlockPtrByIndex(b1[i]);
    auto ex = esdf.getBlockPtrByIndex(b1[i]);
    if (std::memcmp(&x->getVoxelByLinearIndex(0), &bulk_t[i]->getVoxelByLinearIndex(0), x->num_voxels() * sizeof(TsdfVoxel)) ||
        x->updated_bits != bulk_t[i]->updated_bits || x->has_data_flag != bulk_t[i]->has_data_flag)
      return 14;
    if (std::memcmp(&ex->getVoxelByLinearIndex(0), &bulk_e[i]->getVoxelByLinearIndex(0), ex->num_voxels() * sizeof(EsdfVoxel)) ||
        ex->updated_bits != bulk_e[i]->updated_bits)
      return 15;
  }
  // addNewRobotPosition: clear + occupied spheres allocate ESDF-only blocks
  const size_t before = esdf.getNumberOfAllocatedBlocks();
  esdf_integrator.addNewRobotPosition(Point{0.f, 0.f, 0.f});
  esdf_integrator.updateFromTsdfLayer(true);
  std::printf("robot sphere: esdf blocks %zu -> %zu (tsdf %zu)\n", before, esdf.getNumberOfAllocatedBlocks(),
              tsdf.getNumberOfAllocatedBlocks());
  if (esdf.getNumberOfAllocatedBlocks() <= before || tsdf.getNumberOfAllocatedBlocks() != b1.size()) return 16;
  // Multi-GPU host path (vbx_sharded.hpp over libvbx_shard.so), one rank with a real RCCL communicator: the
  // cloud in two ray bands -> delta map -> sparse all-to-all-v to the owner (this rank) -> persistent map.
  {
    DeviceMap persistent(voxel, 16, 2048, 0), delta(voxel, 16, 2048, 0);
    uint8_t id[VBX_SHARD_ID_BYTES];
    ShardedTsdfIntegrator::createCommId(id);
    ShardedTsdfIntegrator sharded(TsdfIntegratorType::kFast, config, &persistent, &delta, 0, 1, id, 0);
    // page-locked host memory is device-accessible on ROCm: stands in for the sensor driver's device buffer
    const size_t n = points.size();
    float* d_pts = static_cast<float*>(vbx_host_alloc(n * 12));
    uint8_t* d_col = static_cast<uint8_t*>(vbx_host_alloc(n * 4));
    if (!d_pts || !d_col) return 30;
    std::memcpy(d_pts, &points[0].x, n * 12);
    std::memcpy(d_col, &colors[0].r, n * 4);
    for (int step = 0; step < 2; ++step) {
      sharded.beginStep();
      sharded.integratePointCloudDevice(T_G_C, d_pts, d_col, n / 2);
      sharded.integratePointCloudDevice(T_G_C, d_pts + 3 * (n / 2), d_col + 4 * (n / 2), n - n / 2);
      sharded.endStep();
    }
    size_t nb = 0;
    persistent.check(vbx_num_blocks(persistent.ctx(), VBX_LAYER_TSDF, &nb), "vbx_num_blocks");
    const vbx_shard_stats st = sharded.stats();
    std::printf("sharded blocks=%zu steps=%llu sent=%llu received=%llu\n", nb, (unsigned long long)st.steps,
                (unsigned long long)st.sent_blocks, (unsigned long long)st.received_blocks);
    vbx_host_free(d_pts);
    vbx_host_free(d_col);
    if (nb == 0 || st.steps != 2 || st.sent_blocks != st.received_blocks || nb * 2 != st.sent_blocks) return 31;
  }
  std::printf("shim OK\n");
  return 0;
}
