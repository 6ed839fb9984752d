// Host-side C++ mirror of the reference's mesher interface over the C-ABI (include/vbx_hip.h):
//   /root/reference/voxblox/include/voxblox/mesh/mesh.h             Mesh            (:35-162)
//   /root/reference/voxblox/include/voxblox/mesh/mesh_layer.h       MeshLayer       (:23-312, the
//                                                                   members the integrator and
//                                                                   voxblox_ros's mesh publishers use)
//   /root/reference/voxblox/include/voxblox/mesh/mesh_integrator.h  MeshIntegratorConfig (:47-66),
//                                                                   MeshIntegrator<VoxelType> (:72-412)
// Same names, member meaning and defaults, so voxblox_ros/src/tsdf_server.cc:92-106 (construction)
// and :420-470 (updateMesh / generateMesh) compile against these with only the namespace changed.
// The marching cubes run on the device (vbx_mesh_generate); the MeshLayer is the host structure it
// is in the reference, and generateMesh() stores each re-meshed block's vertices in it.
#ifndef VBX_MESH_HPP_
#define VBX_MESH_HPP_

#include <cstring>
#include <unordered_map>

#include "vbx_integrators.hpp"

namespace vbx_host {

typedef size_t VertexIndex;  // core/common.h:71
typedef std::vector<VertexIndex> VertexIndexList;

struct Mesh {  // mesh/mesh.h:35-162
  typedef std::shared_ptr<Mesh> Ptr;
  typedef std::shared_ptr<const Mesh> ConstPtr;
  static constexpr FloatingPoint kInvalidBlockSize = -1.0;
  Mesh() : block_size(kInvalidBlockSize), origin{0, 0, 0}, updated(false) {}
  Mesh(FloatingPoint _block_size, const Point& _origin) : block_size(_block_size), origin(_origin), updated(false) {
    VBX_CHECK(block_size > 0.0, "block_size");
  }
  bool hasVertices() const { return !vertices.empty(); }
  bool hasNormals() const { return !normals.empty(); }
  bool hasColors() const { return !colors.empty(); }
  bool hasTriangles() const { return !indices.empty(); }
  size_t size() const { return vertices.size(); }
  void clear() {
    vertices.clear();
    normals.clear();
    colors.clear();
    indices.clear();
  }
  Pointcloud vertices;
  VertexIndexList indices;
  Pointcloud normals;
  Colors colors;
  FloatingPoint block_size;
  Point origin;
  bool updated;
};

class MeshLayer {  // mesh/mesh_layer.h:23-312
 public:
  typedef std::shared_ptr<MeshLayer> Ptr;
  explicit MeshLayer(FloatingPoint block_size) : block_size_(block_size), block_size_inv_(1.0 / block_size) {}

  Mesh::Ptr getMeshPtrByIndex(const BlockIndex& index) {  // :66-74
    auto it = mesh_map_.find(key(index));
    return it == mesh_map_.end() ? Mesh::Ptr() : it->second;
  }
  Mesh::ConstPtr getMeshPtrByIndex(const BlockIndex& index) const {
    auto it = mesh_map_.find(key(index));
    return it == mesh_map_.end() ? Mesh::ConstPtr() : it->second;
  }
  Mesh& getMeshByIndex(const BlockIndex& index) {  // :45-52: LOG(FATAL) on a missing mesh
    Mesh::Ptr m = getMeshPtrByIndex(index);
    VBX_CHECK(m != nullptr, "Accessed unallocated mesh");
    return *m;
  }
  Mesh::Ptr allocateMeshPtrByIndex(const BlockIndex& index) {  // :80-87, :112-122
    auto it = mesh_map_.find(key(index));
    if (it != mesh_map_.end()) return it->second;
    Mesh::Ptr m = std::make_shared<Mesh>(
        block_size_, Point{static_cast<float>(index.x) * block_size_, static_cast<float>(index.y) * block_size_,
                           static_cast<float>(index.z) * block_size_});
    mesh_map_.emplace(key(index), m);
    index_of_[key(index)] = index;
    return m;
  }
  void removeMesh(const BlockIndex& index) {  // :128
    mesh_map_.erase(key(index));
    index_of_.erase(key(index));
  }
  void clearDistantMesh(const Point& center, const double max_distance) {  // :134-144
    for (auto& kv : mesh_map_) {
      const Point& o = kv.second->origin;
      const float dx = o.x - center.x, dy = o.y - center.y, dz = o.z - center.z;
      if (dx * dx + (dy * dy + dz * dz) > max_distance * max_distance) {
        kv.second->clear();
        kv.second->updated = true;
      }
    }
  }
  void getAllAllocatedMeshes(BlockIndexList* meshes) const {  // :146-154
    meshes->clear();
    meshes->reserve(mesh_map_.size());
    for (const auto& kv : index_of_) meshes->push_back(kv.second);
  }
  void getAllUpdatedMeshes(BlockIndexList* meshes) const {  // :156-164
    meshes->clear();
    for (const auto& kv : mesh_map_)
      if (kv.second->updated) meshes->push_back(index_of_.at(kv.first));
  }
  void clear() {  // :275
    mesh_map_.clear();
    index_of_.clear();
  }
  size_t getNumberOfAllocatedMeshes() const { return mesh_map_.size(); }  // :277
  FloatingPoint block_size() const { return block_size_; }
  FloatingPoint block_size_inv() const { return block_size_inv_; }

 private:
  static uint64_t key(const BlockIndex& i) {
    return (static_cast<uint64_t>(static_cast<uint32_t>(i.z) & 0x1FFFFF) << 42) |
           (static_cast<uint64_t>(static_cast<uint32_t>(i.y) & 0x1FFFFF) << 21) |
           static_cast<uint64_t>(static_cast<uint32_t>(i.x) & 0x1FFFFF);
  }
  FloatingPoint block_size_;
  FloatingPoint block_size_inv_;
  std::unordered_map<uint64_t, Mesh::Ptr> mesh_map_;
  std::unordered_map<uint64_t, BlockIndex> index_of_;
};

struct MeshIntegratorConfig {  // mesh/mesh_integrator.h:47-66
  bool use_color = true;
  float min_weight = 1e-4;
  size_t integrator_threads = std::thread::hardware_concurrency();  // kept for source compatibility
};

template <typename VoxelType>
class MeshIntegrator;

template <>
class MeshIntegrator<TsdfVoxel> {  // mesh/mesh_integrator.h:72-412
 public:
  // :93-110 — mutable layer: may clear Update::kMesh
  MeshIntegrator(const MeshIntegratorConfig& config, Layer<TsdfVoxel>* sdf_layer, MeshLayer* mesh_layer)
      : config_(config), sdf_layer_mutable_(sdf_layer), sdf_layer_const_(sdf_layer), mesh_layer_(mesh_layer) {
    VBX_CHECK(sdf_layer != nullptr, "sdf_layer");
    VBX_CHECK(mesh_layer != nullptr, "mesh_layer");
  }
  // :116-137 — const layer: generateMesh(…, clear_updated_flag = true) is a CHECK failure
  MeshIntegrator(const MeshIntegratorConfig& config, const Layer<TsdfVoxel>& sdf_layer, MeshLayer* mesh_layer)
      : config_(config), sdf_layer_mutable_(nullptr), sdf_layer_const_(&sdf_layer), mesh_layer_(mesh_layer) {
    VBX_CHECK(mesh_layer != nullptr, "mesh_layer");
  }

  // :140-172
  void generateMesh(bool only_mesh_updated_blocks, bool clear_updated_flag) {
    VBX_CHECK(!clear_updated_flag || (sdf_layer_mutable_ != nullptr),
              "If you would like to modify the updated flag in the blocks, please use the constructor that "
              "provides a non-const link to the sdf layer!");
    const DeviceMap& m = *sdf_layer_const_->map();
    vbx_mesh_cfg c;
    vbx_mesh_cfg_default(&c);
    c.use_color = config_.use_color ? 1 : 0;
    c.min_weight = config_.min_weight;
    size_t n_blocks = 0, n_vertices = 0;
    m.check(vbx_mesh_generate(m.ctx(), &c, only_mesh_updated_blocks ? 1 : 0, clear_updated_flag ? 1 : 0, &n_blocks,
                              &n_vertices),
            "vbx_mesh_generate");
    idx_.resize(n_blocks * 3);
    off_.resize(n_blocks + 1);
    size_t n = 0;
    m.check(vbx_mesh_blocks(m.ctx(), idx_.data(), off_.data(), n_blocks, &n), "vbx_mesh_blocks");
    verts_.resize(n_vertices);
    normals_.resize(n_vertices);
    if (config_.use_color) colors_.resize(n_vertices);
    m.check(vbx_mesh_download(m.ctx(), n_vertices ? &verts_[0].x : nullptr, n_vertices ? &normals_[0].x : nullptr,
                              (config_.use_color && n_vertices) ? &colors_[0].r : nullptr, n_vertices),
            "vbx_mesh_download");
    for (size_t b = 0; b < n_blocks; ++b) {
      // allocateMeshPtrByIndex (:156-158) + updateMeshForBlock (:250-270)
      Mesh::Ptr mesh = mesh_layer_->allocateMeshPtrByIndex(BlockIndex{idx_[3 * b], idx_[3 * b + 1], idx_[3 * b + 2]});
      const size_t a = off_[b], e = off_[b + 1];
      mesh->clear();
      mesh->vertices.assign(verts_.begin() + a, verts_.begin() + e);
      mesh->normals.assign(normals_.begin() + a, normals_.begin() + e);
      if (config_.use_color) mesh->colors.assign(colors_.begin() + a, colors_.begin() + e);
      mesh->indices.resize(e - a);
      for (size_t i = 0; i < e - a; ++i) mesh->indices[i] = i;  // marching_cubes.h:94-96
      mesh->updated = true;
    }
  }

 private:
  MeshIntegratorConfig config_;
  Layer<TsdfVoxel>* sdf_layer_mutable_;
  const Layer<TsdfVoxel>* sdf_layer_const_;
  MeshLayer* mesh_layer_;
  std::vector<int32_t> idx_;
  std::vector<uint64_t> off_;
  Pointcloud verts_, normals_;
  Colors colors_;
};

}  // namespace vbx_host

#endif  // VBX_MESH_HPP_
