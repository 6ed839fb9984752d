// ORACLE — TEST INFRASTRUCTURE ONLY (see vbx_oracle.h).
//
// CPU restatement of the reference's incremental mesher:
//   voxblox/include/voxblox/mesh/mesh.h            Mesh
//   voxblox/include/voxblox/mesh/mesh_layer.h      MeshLayer (allocate / get by index)
//   voxblox/include/voxblox/mesh/marching_cubes.h  MarchingCubes
//   voxblox/include/voxblox/mesh/mesh_integrator.h MeshIntegrator<TsdfVoxel>
//   voxblox/include/voxblox/utils/meshing_utils.h  getSdfIfValid / getColorIfValid
// Each function cites the lines it follows.  Eigen call sites restated as in vbx_core.hpp
// (cross, normalized() = v / sqrt(c0 + (c1 + c2)) guarded by > 0) — unpinned third-party
// arithmetic, shared with ref_shims/Eigen/Core.
#pragma once
#include <cmath>
#include <cstdint>
#include <memory>
#include <unordered_map>
#include <vector>

#include "vbx_core.hpp"
#include "vbx_mc_table.hpp"

namespace orc {

// mesh.h:35-162 (the fields the integrator writes)
struct Mesh {
  std::vector<Vec3f> vertices;
  std::vector<size_t> indices;  // VertexIndex = size_t (common.h:71)
  std::vector<Vec3f> normals;
  std::vector<Color> colors;
  float block_size = -1.0f;
  Vec3f origin{0, 0, 0};
  bool updated = false;
  void clear() {  // mesh.h:76-81
    vertices.clear();
    normals.clear();
    colors.clear();
    indices.clear();
  }
};

// mesh_layer.h:23-133
struct MeshLayer {
  explicit MeshLayer(float block_size) : block_size(block_size), block_size_inv(1.0 / block_size) {}
  std::shared_ptr<Mesh> getMeshPtrByIndex(const Idx3& i) const {
    auto it = mesh_map.find(i);
    return it == mesh_map.end() ? nullptr : it->second;
  }
  // mesh_layer.h:77-84, :112-122
  std::shared_ptr<Mesh> allocateMeshPtrByIndex(const Idx3& i) {
    auto it = mesh_map.find(i);
    if (it != mesh_map.end()) return it->second;
    auto m = std::make_shared<Mesh>();
    m->block_size = block_size;
    m->origin = {static_cast<float>(i.x) * block_size, static_cast<float>(i.y) * block_size,
                 static_cast<float>(i.z) * block_size};
    mesh_map.emplace(i, m);
    return m;
  }
  float block_size, block_size_inv;
  std::unordered_map<Idx3, std::shared_ptr<Mesh>, AnyIndexHasher> mesh_map;
};

// marching_cubes.h:35-163
struct MarchingCubes {
  // marching_cubes.h:148-161
  static Vec3f interpolateVertex(const Vec3f& vertex1, const Vec3f& vertex2, float sdf1, float sdf2) {
    constexpr float kMinSdfDifference = 1e-6;
    const float sdf_diff = sdf1 - sdf2;
    if (std::abs(sdf_diff) >= kMinSdfDifference) {
      const float t = sdf1 / sdf_diff;
      return vertex1 + (vertex2 - vertex1) * t;
    }
    // 0.5 * (vertex1 + vertex2): the double literal becomes the matrix scalar type
    return (vertex1 + vertex2) * 0.5f;
  }
  // marching_cubes.h:113-123
  static int calculateVertexConfiguration(const float sdf[8]) {
    int index = 0;
    for (int i = 0; i < 8; ++i) index |= (sdf[i] < 0 ? (1 << i) : 0);
    return index;
  }
  // marching_cubes.h:125-141: only edges with a zero crossing are written; the others keep
  // whatever the (uninitialised) matrix held and are never read, because the case table only
  // references crossing edges
  static void interpolateEdgeVertices(const Vec3f coords[8], const float sdf[8], Vec3f edge_coords[12]) {
    for (int i = 0; i < 12; ++i) {
      const int e0 = orc_mc::kMcEdgeCorners[i][0], e1 = orc_mc::kMcEdgeCorners[i][1];
      if ((sdf[e0] < 0 && sdf[e1] >= 0) || (sdf[e0] >= 0 && sdf[e1] < 0))
        edge_coords[i] = interpolateVertex(coords[e0], coords[e1], sdf[e0], sdf[e1]);
    }
  }
  // marching_cubes.h:70-111
  static void meshCube(const Vec3f coords[8], const float sdf[8], size_t* next_index, Mesh* mesh) {
    const int index = calculateVertexConfiguration(sdf);
    if (index == 0) return;
    Vec3f edge[12];
    for (auto& e : edge) e = Vec3f{0, 0, 0};
    interpolateEdgeVertices(coords, sdf, edge);
    const uint64_t row = orc_mc::kMcTriTable[index];
    const int n_tri = static_cast<int>(row >> 60);
    for (int t = 0; t < n_tri; ++t) {
      const int c = 3 * t;
      mesh->vertices.push_back(edge[(row >> (4 * (c + 2))) & 15]);
      mesh->vertices.push_back(edge[(row >> (4 * (c + 1))) & 15]);
      mesh->vertices.push_back(edge[(row >> (4 * c)) & 15]);
      mesh->indices.push_back(*next_index);
      mesh->indices.push_back(*next_index + 1);
      mesh->indices.push_back(*next_index + 2);
      const Vec3f& p0 = mesh->vertices[*next_index];
      const Vec3f& p1 = mesh->vertices[*next_index + 1];
      const Vec3f& p2 = mesh->vertices[*next_index + 2];
      const Vec3f px = p1 - p0, py = p2 - p0;
      const Vec3f n = normalized(cross(px, py));
      mesh->normals.push_back(n);
      mesh->normals.push_back(n);
      mesh->normals.push_back(n);
      *next_index += 3;
    }
  }
};

struct MeshIntegratorConfig {  // mesh_integrator.h:47-66
  bool use_color = true;
  float min_weight = 1e-4;
  size_t integrator_threads = 1;
};

// mesh_integrator.h:72-412, VoxelType = TsdfVoxel
class MeshIntegrator {
 public:
  MeshIntegrator(const MeshIntegratorConfig& config, Layer<TsdfVoxel>* sdf_layer, MeshLayer* mesh_layer)
      : config_(config), sdf_layer_(sdf_layer), mesh_layer_(mesh_layer) {
    voxel_size_ = sdf_layer->voxel_size;          // :77-85
    voxels_per_side_ = sdf_layer->voxels_per_side;
  }

  // mesh_integrator.h:142-172.  Blocks are independent, so the worker threads of the reference
  // (MixedThreadSafeIndex over the list) cannot change the result; run in list order.
  void generateMesh(bool only_mesh_updated_blocks, bool clear_updated_flag) {
    std::vector<Idx3> all_tsdf_blocks;
    if (only_mesh_updated_blocks) sdf_layer_->getAllUpdatedBlocks(1 /* Update::kMesh */, &all_tsdf_blocks);
    else sdf_layer_->getAllAllocatedBlocks(&all_tsdf_blocks);
    for (const Idx3& b : all_tsdf_blocks) mesh_layer_->allocateMeshPtrByIndex(b);
    for (const Idx3& b : all_tsdf_blocks) {  // :174-195
      updateMeshForBlock(b);
      if (clear_updated_flag) sdf_layer_->getBlockPtrByIndex(b)->updated &= static_cast<uint8_t>(~2u);
    }
  }

  // mesh_integrator.h:250-270
  void updateMeshForBlock(const Idx3& block_index) {
    std::shared_ptr<Mesh> mesh = mesh_layer_->getMeshPtrByIndex(block_index);
    mesh->clear();
    auto block = sdf_layer_->getBlockPtrByIndex(block_index);
    if (!block) return;
    extractBlockMesh(*block, mesh.get());
    if (config_.use_color) updateMeshColor(*block, mesh.get());
    mesh->updated = true;
  }

 private:
  // block.h:90-92
  static Vec3f coordsFromVoxelIndex(const Block<TsdfVoxel>& b, const Idx3& v) {
    return b.origin + centerPointFromGridIndex(v, b.voxel_size);
  }
  // mesh_integrator.h:197-248 — note the interior loop runs x outermost, z innermost
  void extractBlockMesh(const Block<TsdfVoxel>& block, Mesh* mesh) {
    const int vps = static_cast<int>(block.voxels_per_side);
    size_t next_mesh_index = 0;
    Idx3 v;
    for (v.x = 0; v.x < vps - 1; ++v.x)
      for (v.y = 0; v.y < vps - 1; ++v.y)
        for (v.z = 0; v.z < vps - 1; ++v.z)
          extractMeshInsideBlock(block, v, coordsFromVoxelIndex(block, v), &next_mesh_index, mesh);
    v.x = vps - 1;  // max X plane
    for (v.z = 0; v.z < vps; v.z++)
      for (v.y = 0; v.y < vps; v.y++)
        extractMeshOnBorder(block, v, coordsFromVoxelIndex(block, v), &next_mesh_index, mesh);
    v.y = vps - 1;  // max Y plane
    for (v.z = 0; v.z < vps; v.z++)
      for (v.x = 0; v.x < vps - 1; v.x++)
        extractMeshOnBorder(block, v, coordsFromVoxelIndex(block, v), &next_mesh_index, mesh);
    v.z = vps - 1;  // max Z plane
    for (v.y = 0; v.y < vps - 1; v.y++)
      for (v.x = 0; v.x < vps - 1; v.x++)
        extractMeshOnBorder(block, v, coordsFromVoxelIndex(block, v), &next_mesh_index, mesh);
  }
  // meshing_utils.h:15-24
  static bool getSdfIfValid(const TsdfVoxel& voxel, float min_weight, float* sdf) {
    if (voxel.weight <= min_weight) return false;
    *sdf = voxel.distance;
    return true;
  }
  Vec3f cornerCoord(const Vec3f& coords, int i) const {
    // cube_index_offsets_.cast<float>() * voxel_size_, then coords + column (:277-278, :290)
    return coords + Vec3f{static_cast<float>(orc_mc::kMcCornerOffset[i][0]) * voxel_size_,
                          static_cast<float>(orc_mc::kMcCornerOffset[i][1]) * voxel_size_,
                          static_cast<float>(orc_mc::kMcCornerOffset[i][2]) * voxel_size_};
  }
  // mesh_integrator.h:272-300
  void extractMeshInsideBlock(const Block<TsdfVoxel>& block, const Idx3& index, const Vec3f& coords,
                              size_t* next_mesh_index, Mesh* mesh) {
    Vec3f corner_coords[8];
    float corner_sdf[8];
    for (int i = 0; i < 8; ++i) {
      const Idx3 c{index.x + orc_mc::kMcCornerOffset[i][0], index.y + orc_mc::kMcCornerOffset[i][1],
                   index.z + orc_mc::kMcCornerOffset[i][2]};
      if (!getSdfIfValid(block.voxels[block.linearIndex(c)], config_.min_weight, &corner_sdf[i])) return;
      corner_coords[i] = cornerCoord(coords, i);
    }
    MarchingCubes::meshCube(corner_coords, corner_sdf, next_mesh_index, mesh);
  }
  // mesh_integrator.h:302-370
  void extractMeshOnBorder(const Block<TsdfVoxel>& block, const Idx3& index, const Vec3f& coords,
                           size_t* next_mesh_index, Mesh* mesh) {
    Vec3f corner_coords[8];
    float corner_sdf[8];
    const int vps = static_cast<int>(voxels_per_side_);
    for (int i = 0; i < 8; ++i) {
      Idx3 c{index.x + orc_mc::kMcCornerOffset[i][0], index.y + orc_mc::kMcCornerOffset[i][1],
             index.z + orc_mc::kMcCornerOffset[i][2]};
      if (block.isValidVoxelIndex(c)) {
        if (!getSdfIfValid(block.voxels[block.linearIndex(c)], config_.min_weight, &corner_sdf[i])) return;
      } else {
        Idx3 off{0, 0, 0};
        int* cp[3] = {&c.x, &c.y, &c.z};
        int* op[3] = {&off.x, &off.y, &off.z};
        for (int j = 0; j < 3; ++j) {
          if (*cp[j] < 0) { *op[j] = -1; *cp[j] += vps; }
          else if (*cp[j] >= vps) { *op[j] = 1; *cp[j] -= vps; }
        }
        const Idx3 bi = block.block_index();
        const Idx3 neighbor_index{bi.x + off.x, bi.y + off.y, bi.z + off.z};
        auto nb = sdf_layer_->getBlockPtrByIndex(neighbor_index);
        if (!nb) return;
        if (!getSdfIfValid(nb->voxels[nb->linearIndex(c)], config_.min_weight, &corner_sdf[i])) return;
      }
      corner_coords[i] = cornerCoord(coords, i);
    }
    MarchingCubes::meshCube(corner_coords, corner_sdf, next_mesh_index, mesh);
  }
  // mesh_integrator.h:372-392 (+ block.h:65-70, block_inl.h:30-41, layer.h:105-108)
  void updateMeshColor(const Block<TsdfVoxel>& block, Mesh* mesh) {
    mesh->colors.clear();
    mesh->colors.resize(mesh->indices.size());
    for (size_t i = 0; i < mesh->vertices.size(); i++) {
      const Vec3f& vertex = mesh->vertices[i];
      const Idx3 voxel_index = gridIndexFromPointI(vertex - block.origin, block.voxel_size_inv);
      const TsdfVoxel* voxel;
      if (block.isValidVoxelIndex(voxel_index)) {
        voxel = &block.voxels[block.linearIndex(voxel_index)];
      } else {
        auto nb = sdf_layer_->getBlockPtrByIndex(gridIndexFromPointI(vertex, sdf_layer_->block_size_inv));
        if (!nb) continue;  // the reference dereferences a null pointer here; cannot happen for
                            // a vertex between observed corners (their blocks exist)
        const int max_value = static_cast<int>(nb->voxels_per_side) - 1;
        Idx3 t = gridIndexFromPointI(vertex - nb->origin, nb->voxel_size_inv);  // block_inl.h:30-41
        t = {std::max(std::min(t.x, max_value), 0), std::max(std::min(t.y, max_value), 0),
             std::max(std::min(t.z, max_value), 0)};
        voxel = &nb->voxels[nb->linearIndex(t)];
      }
      if (voxel->weight > config_.min_weight) mesh->colors[i] = voxel->color;  // meshing_utils.h:43-52
    }
  }

  MeshIntegratorConfig config_;
  Layer<TsdfVoxel>* sdf_layer_;
  MeshLayer* mesh_layer_;
  float voxel_size_;
  size_t voxels_per_side_;
};

}  // namespace orc
