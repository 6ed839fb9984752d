// Host-side C++ mirror of the reference's integrator interface over the C-ABI (include/vbx_hip.h).
//
// Same class names, member names, argument meaning, defaults and error behaviour as
//   /root/reference/voxblox/include/voxblox/integrator/tsdf_integrator.h  (:51-341)
//   /root/reference/voxblox/include/voxblox/integrator/esdf_integrator.h  (:24-179)
//   /root/reference/voxblox/include/voxblox/core/layer.h                  (:24-296, the members
//                                                                          the integrators' callers use)
// so code written against voxblox's integrators (voxblox_ros/src/tsdf_server.cc:92-106, 407-414;
// esdf_server.cc:192-197; test/test_sdf_integrators.cc) compiles against these with only the
// namespace changed.  Header-only, depends on nothing but the C-ABI: Eigen / minkindr / glog are
// not in this image, so Point, Color and Transformation are PODs with the reference's layout
// (Point = 3 floats, 12-byte stride like Eigen::Vector3f inside AlignedVector; Color = 4 bytes).
// INTEGRATION.md shows the same binding written against the real voxblox types.
//
// The Layer lives in HBM.  Layer<V> here is a handle onto it; getBlockByIndex() downloads a
// host copy of one block in the reference's AoS voxel layout.
#ifndef VBX_INTEGRATORS_HPP_
#define VBX_INTEGRATORS_HPP_

#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vbx_hip.h"

namespace vbx_host {

// glog CHECK / LOG(FATAL) convention of the reference: print and abort.
#define VBX_CHECK(cond, msg)                                                        \
  do {                                                                              \
    if (!(cond)) {                                                                  \
      std::fprintf(stderr, "Check failed: %s %s (%s:%d)\n", #cond, msg, __FILE__, __LINE__); \
      std::abort();                                                                 \
    }                                                                               \
  } while (0)

typedef float FloatingPoint;  // core/common.h:41
struct Point { float x, y, z; };
struct Color { uint8_t r = 0, g = 0, b = 0, a = 0; };
typedef std::vector<Point> Pointcloud;  // core/common.h:67
typedef std::vector<Color> Colors;      // core/common.h:68
struct BlockIndex { int32_t x, y, z; };
typedef std::vector<BlockIndex> BlockIndexList;
static_assert(sizeof(Point) == 12 && sizeof(Color) == 4, "reference layouts");

// kindr::minimal::QuatTransformationTemplate<float> (core/common.h:77-78): the subset used on
// the integration path.
class Transformation {
 public:
  Transformation() : position_{0, 0, 0}, quat_wxyz_{1, 0, 0, 0} {}
  Transformation(const Point& position, const std::array<float, 4>& quat_wxyz)
      : position_(position), quat_wxyz_(quat_wxyz) {}
  const Point& getPosition() const { return position_; }
  const std::array<float, 4>& getRotationWxyz() const { return quat_wxyz_; }

 private:
  Point position_;
  std::array<float, 4> quat_wxyz_;
};

struct TsdfVoxel {  // core/voxel.h:12-16
  float distance = 0.0f;
  float weight = 0.0f;
  Color color;
};
struct EsdfVoxel {  // core/voxel.h:18-37
  float distance = 0.0f;
  bool observed = false;
  bool hallucinated = false;
  bool in_queue = false;
  bool fixed = false;
  int32_t parent[3] = {0, 0, 0};
};
static_assert(sizeof(TsdfVoxel) == 12 && sizeof(EsdfVoxel) == 20, "reference layouts");

namespace Update {  // core/block.h:15-18
enum Status { kMap = 0, kMesh = 1, kEsdf = 2, kCount = 3 };
}

template <typename VoxelType> struct LayerId;
template <> struct LayerId<TsdfVoxel> { static constexpr int value = VBX_LAYER_TSDF; };
template <> struct LayerId<EsdfVoxel> { static constexpr int value = VBX_LAYER_ESDF; };

// Host copy of one block (core/block.h:23-215: the accessors callers use).
template <typename VoxelType>
class Block {
 public:
  Block(size_t voxels_per_side, FloatingPoint voxel_size)
      : voxels_per_side_(voxels_per_side), voxel_size_(voxel_size),
        voxels_(voxels_per_side * voxels_per_side * voxels_per_side) {}
  size_t num_voxels() const { return voxels_.size(); }
  size_t voxels_per_side() const { return voxels_per_side_; }
  const VoxelType& getVoxelByLinearIndex(size_t i) const { return voxels_[i]; }
  VoxelType& getVoxelByLinearIndex(size_t i) { return voxels_[i]; }
  bool updated(Update::Status bit) const { return (updated_bits >> bit) & 1; }
  bool has_data() const { return has_data_flag != 0; }
  VoxelType* data() { return voxels_.data(); }
  uint8_t updated_bits = 0;
  uint8_t has_data_flag = 0;

 private:
  size_t voxels_per_side_;
  FloatingPoint voxel_size_;
  std::vector<VoxelType> voxels_;
};

// The HBM-resident map: one TSDF layer and one ESDF layer of equal geometry share a handle
// ("Block indices are the same across all layers", esdf_integrator.cc:144).
class DeviceMap {
 public:
  DeviceMap(FloatingPoint voxel_size, size_t voxels_per_side, uint32_t max_blocks = 0, int device = 0) {
    VBX_CHECK(voxel_size > 0.0f, "voxel_size");  // layer.h:36
    vbx_map_cfg cfg{voxel_size, static_cast<uint32_t>(voxels_per_side), max_blocks};
    ctx_ = vbx_create(&cfg, device);
    if (!ctx_) {
      std::fprintf(stderr, "vbx_create failed: %s\n", vbx_last_error(nullptr));
      std::abort();
    }
    voxel_size_ = voxel_size;
    voxels_per_side_ = voxels_per_side;
  }
  ~DeviceMap() { vbx_destroy(ctx_); }
  DeviceMap(const DeviceMap&) = delete;
  DeviceMap& operator=(const DeviceMap&) = delete;
  vbx_ctx* ctx() const { return ctx_; }
  void check(int rc, const char* what) const {
    if (rc != VBX_OK) {
      std::fprintf(stderr, "%s failed (%d): %s\n", what, rc, vbx_last_error(ctx_));
      std::abort();
    }
  }
  FloatingPoint voxel_size() const { return voxel_size_; }
  size_t voxels_per_side() const { return voxels_per_side_; }

 private:
  vbx_ctx* ctx_ = nullptr;
  FloatingPoint voxel_size_ = 0;
  size_t voxels_per_side_ = 0;
};

// Layer<VoxelType> (core/layer.h:24-296): a typed view of the device map.
template <typename VoxelType>
class Layer {
 public:
  typedef std::shared_ptr<Layer> Ptr;
  typedef Block<VoxelType> BlockType;
  explicit Layer(std::shared_ptr<DeviceMap> map) : map_(std::move(map)) { VBX_CHECK(map_ != nullptr, "map"); }
  // Layer(voxel_size, voxels_per_side), layer.h:34-44: a fresh map of its own.
  Layer(FloatingPoint voxel_size, size_t voxels_per_side)
      : map_(std::make_shared<DeviceMap>(voxel_size, voxels_per_side)) {}

  FloatingPoint voxel_size() const { return map_->voxel_size(); }
  size_t voxels_per_side() const { return map_->voxels_per_side(); }
  FloatingPoint block_size() const { return map_->voxel_size() * map_->voxels_per_side(); }
  const std::shared_ptr<DeviceMap>& map() const { return map_; }

  size_t getNumberOfAllocatedBlocks() const {  // layer.h:205
    size_t n = 0;
    map_->check(vbx_num_blocks(map_->ctx(), kId, &n), "vbx_num_blocks");
    return n;
  }
  void getAllAllocatedBlocks(BlockIndexList* blocks) const {  // layer.h:184-192
    VBX_CHECK(blocks != nullptr, "blocks");
    size_t n = getNumberOfAllocatedBlocks();
    blocks->resize(n);
    map_->check(vbx_block_indices(map_->ctx(), kId, n ? &(*blocks)[0].x : nullptr, n, &n), "vbx_block_indices");
    blocks->resize(n);
  }
  void getAllUpdatedBlocks(Update::Status bit, BlockIndexList* blocks) const {  // layer.h:194-203
    VBX_CHECK(blocks != nullptr, "blocks");
    size_t n = 0;
    map_->check(vbx_blocks_updated(map_->ctx(), kId, 1 << bit, nullptr, 0, &n), "vbx_blocks_updated");
    blocks->resize(n);
    map_->check(vbx_blocks_updated(map_->ctx(), kId, 1 << bit, n ? &(*blocks)[0].x : nullptr, n, &n),
                "vbx_blocks_updated");
    blocks->resize(n);
  }
  // getBlockPtrByIndex (layer.h:72-89): nullptr when the block is not allocated.
  std::shared_ptr<BlockType> getBlockPtrByIndex(const BlockIndex& index) const {
    auto b = std::make_shared<BlockType>(voxels_per_side(), voxel_size());
    const int rc = vbx_block_download(map_->ctx(), kId, &index.x, b->data(), &b->updated_bits, &b->has_data_flag);
    if (rc == VBX_ERR_INVALID) return nullptr;
    map_->check(rc, "vbx_block_download");
    return b;
  }
  // Bulk form of the loop `for (idx : list) getBlockPtrByIndex(idx)` — one device pack + one copy
  // (vbx_blocks_download); what a per-frame host mirror of getAllUpdatedBlocks(kMap) should use.
  void getBlocksByIndex(const BlockIndexList& list, std::vector<std::shared_ptr<BlockType>>* out) const {
    VBX_CHECK(out != nullptr, "out");
    const size_t n = list.size();
    out->clear();
    if (n == 0) return;
    const size_t nv = voxels_per_side() * voxels_per_side() * voxels_per_side();
    std::vector<VoxelType> voxels(n * nv);
    std::vector<uint8_t> bits(n), hd(n);
    map_->check(vbx_blocks_download(map_->ctx(), kId, &list[0].x, n, voxels.data(), bits.data(), hd.data()),
                "vbx_blocks_download");
    out->reserve(n);
    for (size_t i = 0; i < n; ++i) {
      auto b = std::make_shared<BlockType>(voxels_per_side(), voxel_size());
      std::copy(voxels.begin() + i * nv, voxels.begin() + (i + 1) * nv, b->data());
      b->updated_bits = bits[i];
      b->has_data_flag = hd[i];
      out->push_back(std::move(b));
    }
  }
  bool hasBlock(const BlockIndex& index) const { return getBlockPtrByIndex(index) != nullptr; }  // layer.h:207
  void insertBlock(const BlockIndex& index, const BlockType& block) {  // load_map path, layer.h:147-157
    map_->check(vbx_block_upload(map_->ctx(), kId, &index.x, &block.getVoxelByLinearIndex(0), block.updated_bits,
                                 block.has_data_flag),
                "vbx_block_upload");
  }
  void removeBlock(const BlockIndex& index) { map_->check(vbx_block_remove(map_->ctx(), kId, &index.x), "vbx_block_remove"); }
  void removeAllBlocks() { map_->check(vbx_clear(map_->ctx(), kId), "vbx_clear"); }
  void removeDistantBlocks(const Point& center, const double max_distance) {  // layer.h:170-182
    map_->check(vbx_remove_distant_blocks(map_->ctx(), kId, &center.x, max_distance), "vbx_remove_distant_blocks");
  }
  void clearUpdatedFlag(Update::Status bit) { map_->check(vbx_clear_updated(map_->ctx(), kId, 1 << bit), "vbx_clear_updated"); }

 private:
  static constexpr int kId = LayerId<VoxelType>::value;
  std::shared_ptr<DeviceMap> map_;
};

// tsdf_integrator.h:30-41
enum class TsdfIntegratorType : int { kSimple = 1, kMerged = 2, kFast = 3 };
static constexpr size_t kNumTsdfIntegratorTypes = 3u;
const std::array<std::string, kNumTsdfIntegratorTypes> kTsdfIntegratorTypeNames = {{"simple", "merged", "fast"}};

class TsdfIntegratorBase {  // tsdf_integrator.h:51-198
 public:
  typedef std::shared_ptr<TsdfIntegratorBase> Ptr;
  struct Config {  // tsdf_integrator.h:56-89, same names and defaults
    float default_truncation_distance = 0.1f;
    float max_weight = 10000.0f;
    bool voxel_carving_enabled = true;
    FloatingPoint min_ray_length_m = 0.1f;
    FloatingPoint max_ray_length_m = 5.0f;
    bool use_const_weight = false;
    bool allow_clear = true;
    bool use_weight_dropoff = true;
    bool use_sparsity_compensation_factor = false;
    float sparsity_compensation_factor = 1.0f;
    size_t integrator_threads = std::thread::hardware_concurrency();  // accepted, ignored on the GPU
    std::string integration_order_mode = "mixed";
    bool enable_anti_grazing = false;
    float start_voxel_subsampling_factor = 2.0f;
    int max_consecutive_ray_collisions = 2;
    int clear_checks_every_n_frames = 1;
    float max_integration_time_s = std::numeric_limits<float>::max();
    // not in the reference: 0 = the reference's unordered_map bundle order (bit-exact), 1 = ascending voxel key (faster)
    int merged_bundle_order = 0;
    // not in the reference: 0 = the reference's approximate observed-voxel set (bit-exact), 1 = exact set (faster)
    int fast_observed_set = 0;
  };

  TsdfIntegratorBase(const Config& config, Layer<TsdfVoxel>* layer) : config_(config) {
    setLayer(layer);
    if (config_.integrator_threads == 0) config_.integrator_threads = 1;                // tsdf_integrator.cc:58-61
    if (config_.allow_clear && !config_.voxel_carving_enabled) config_.allow_clear = false;  // :63-65
  }
  virtual ~TsdfIntegratorBase() = default;

  // tsdf_integrator.h:100-103
  virtual void integratePointCloud(const Transformation& T_G_C, const Pointcloud& points_C, const Colors& colors,
                                   const bool freespace_points = false) = 0;
  const Config& getConfig() const { return config_; }
  /// Config -> the C-ABI's plain struct (same fields, include/vbx_hip.h)
  static vbx_tsdf_cfg toC(const Config& config) {
    vbx_tsdf_cfg c;
    vbx_tsdf_cfg_default(&c);
    c.default_truncation_distance = config.default_truncation_distance;
    c.max_weight = config.max_weight;
    c.voxel_carving_enabled = config.voxel_carving_enabled;
    c.min_ray_length_m = config.min_ray_length_m;
    c.max_ray_length_m = config.max_ray_length_m;
    c.use_const_weight = config.use_const_weight;
    c.allow_clear = config.allow_clear;
    c.use_weight_dropoff = config.use_weight_dropoff;
    c.use_sparsity_compensation_factor = config.use_sparsity_compensation_factor;
    c.sparsity_compensation_factor = config.sparsity_compensation_factor;
    c.integrator_threads = static_cast<int32_t>(config.integrator_threads);
    if (config.integration_order_mode == "mixed") c.integration_order_mode = 0;
    else if (config.integration_order_mode == "sorted") c.integration_order_mode = 1;
    else VBX_CHECK(false, "Unknown integration order mode");  // integrator_utils.cc:12
    c.enable_anti_grazing = config.enable_anti_grazing;
    c.start_voxel_subsampling_factor = config.start_voxel_subsampling_factor;
    c.max_consecutive_ray_collisions = config.max_consecutive_ray_collisions;
    c.clear_checks_every_n_frames = config.clear_checks_every_n_frames;
    c.max_integration_time_s = config.max_integration_time_s;
    c.merged_bundle_order = config.merged_bundle_order;
    c.fast_observed_set = config.fast_observed_set;
    return c;
  }
  void setLayer(Layer<TsdfVoxel>* layer) {  // tsdf_integrator.cc:68-80
    VBX_CHECK(layer != nullptr, "layer");
    layer_ = layer;
  }

 protected:
  void integrate(int kind, const Transformation& T_G_C, const Pointcloud& points_C, const Colors& colors,
                 bool freespace_points) {
    VBX_CHECK(points_C.size() == colors.size(), "points_C.size() == colors.size()");  // tsdf_integrator.cc:247
    const vbx_tsdf_cfg c = toC(config_);
    const DeviceMap& m = *layer_->map();
    m.check(vbx_tsdf_integrate(m.ctx(), kind, &c, &T_G_C.getPosition().x, T_G_C.getRotationWxyz().data(),
                               points_C.empty() ? nullptr : &points_C[0].x,
                               colors.empty() ? nullptr : &colors[0].r, points_C.size(), freespace_points ? 1 : 0),
            "vbx_tsdf_integrate");
  }
  Config config_;
  Layer<TsdfVoxel>* layer_;
};

#define VBX_DEFINE_TSDF_INTEGRATOR(Name, Kind)                                                   \
  class Name : public TsdfIntegratorBase {                                                       \
   public:                                                                                       \
    Name(const Config& config, Layer<TsdfVoxel>* layer) : TsdfIntegratorBase(config, layer) {}   \
    void integratePointCloud(const Transformation& T_G_C, const Pointcloud& points_C,            \
                             const Colors& colors, const bool freespace_points = false) override { \
      integrate(Kind, T_G_C, points_C, colors, freespace_points);                                \
    }                                                                                            \
  }
VBX_DEFINE_TSDF_INTEGRATOR(SimpleTsdfIntegrator, VBX_TSDF_SIMPLE);  // tsdf_integrator.h:215-230
VBX_DEFINE_TSDF_INTEGRATOR(MergedTsdfIntegrator, VBX_TSDF_MERGED);  // tsdf_integrator.h:237-272
VBX_DEFINE_TSDF_INTEGRATOR(FastTsdfIntegrator, VBX_TSDF_FAST);      // tsdf_integrator.h:286-341
#undef VBX_DEFINE_TSDF_INTEGRATOR

class TsdfIntegratorFactory {  // tsdf_integrator.h:201-209, tsdf_integrator.cc:8-46
 public:
  static TsdfIntegratorBase::Ptr create(const std::string& integrator_type_name,
                                        const TsdfIntegratorBase::Config& config, Layer<TsdfVoxel>* layer) {
    VBX_CHECK(!integrator_type_name.empty(), "integrator_type_name");
    int integrator_type = 1;
    for (const std::string& name : kTsdfIntegratorTypeNames) {
      if (integrator_type_name == name) return create(static_cast<TsdfIntegratorType>(integrator_type), config, layer);
      ++integrator_type;
    }
    VBX_CHECK(false, ("Unknown TSDF integrator type: " + integrator_type_name).c_str());
    return TsdfIntegratorBase::Ptr();
  }
  static TsdfIntegratorBase::Ptr create(const TsdfIntegratorType integrator_type,
                                        const TsdfIntegratorBase::Config& config, Layer<TsdfVoxel>* layer) {
    VBX_CHECK(layer != nullptr, "layer");
    switch (integrator_type) {
      case TsdfIntegratorType::kSimple: return TsdfIntegratorBase::Ptr(new SimpleTsdfIntegrator(config, layer));
      case TsdfIntegratorType::kMerged: return TsdfIntegratorBase::Ptr(new MergedTsdfIntegrator(config, layer));
      case TsdfIntegratorType::kFast: return TsdfIntegratorBase::Ptr(new FastTsdfIntegrator(config, layer));
      default: VBX_CHECK(false, "Unknown TSDF integrator type");
    }
    return TsdfIntegratorBase::Ptr();
  }
};

class EsdfIntegrator {  // esdf_integrator.h:24-179
 public:
  struct Config {  // esdf_integrator.h:29-78
    bool full_euclidean_distance = false;
    FloatingPoint max_distance_m = 2.0f;
    FloatingPoint min_distance_m = 0.2f;
    FloatingPoint default_distance_m = 2.0f;
    FloatingPoint min_diff_m = 0.001f;
    float min_weight = 1e-6f;
    int num_buckets = 20;
    bool multi_queue = false;
    bool add_occupied_crust = false;
    FloatingPoint clear_sphere_radius = 1.5f;
    FloatingPoint occupied_sphere_radius = 5.0f;
  };
  EsdfIntegrator(const Config& config, Layer<TsdfVoxel>* tsdf_layer, Layer<EsdfVoxel>* esdf_layer)
      : config_(config), tsdf_layer_(tsdf_layer), esdf_layer_(esdf_layer) {
    VBX_CHECK(tsdf_layer_ != nullptr, "tsdf_layer");  // esdf_integrator.cc:11-12
    VBX_CHECK(esdf_layer_ != nullptr, "esdf_layer");
    VBX_CHECK(tsdf_layer_->map().get() == esdf_layer_->map().get(),
              "TSDF and ESDF layers must be views of one DeviceMap (same geometry, esdf_integrator.cc:17-18)");
  }
  void updateFromTsdfLayer(bool clear_updated_flag) { run(false, clear_updated_flag); }  // esdf_integrator.cc:104-122
  void updateFromTsdfLayerBatch() { run(true, false); }                                   // esdf_integrator.cc:94-102
  void updateFromTsdfBlocks(const BlockIndexList& tsdf_blocks, bool incremental = false) {  // esdf_integrator.cc:124-302
    const vbx_esdf_cfg c = toC();
    const DeviceMap& m = *tsdf_layer_->map();
    m.check(vbx_esdf_update_blocks(m.ctx(), &c, tsdf_blocks.empty() ? nullptr : &tsdf_blocks[0].x, tsdf_blocks.size(),
                                   incremental ? 1 : 0),
            "vbx_esdf_update_blocks");
  }
  void clear() {  // esdf_integrator.h:138-142
    const DeviceMap& m = *tsdf_layer_->map();
    m.check(vbx_esdf_integrator_clear(m.ctx()), "vbx_esdf_integrator_clear");
  }
  void addNewRobotPosition(const Point& position) {                                       // esdf_integrator.cc:25-92
    const vbx_esdf_cfg c = toC();
    const float p[3] = {position.x, position.y, position.z};
    const DeviceMap& m = *tsdf_layer_->map();
    m.check(vbx_esdf_add_new_robot_position(m.ctx(), &c, p), "vbx_esdf_add_new_robot_position");
  }
  float getEsdfMaxDistance() const { return config_.max_distance_m; }
  void setEsdfMaxDistance(float max_distance) {  // esdf_integrator.h:139-144
    config_.max_distance_m = max_distance;
    if (config_.default_distance_m < max_distance) config_.default_distance_m = max_distance;
  }
  bool getFullEuclidean() const { return config_.full_euclidean_distance; }
  void setFullEuclidean(bool full_euclidean) { config_.full_euclidean_distance = full_euclidean; }

 private:
  vbx_esdf_cfg toC() const {
    vbx_esdf_cfg c;
    vbx_esdf_cfg_default(&c);
    c.full_euclidean_distance = config_.full_euclidean_distance;
    c.max_distance_m = config_.max_distance_m;
    c.min_distance_m = config_.min_distance_m;
    c.default_distance_m = config_.default_distance_m;
    c.min_diff_m = config_.min_diff_m;
    c.min_weight = config_.min_weight;
    c.num_buckets = config_.num_buckets;
    c.multi_queue = config_.multi_queue;
    c.add_occupied_crust = config_.add_occupied_crust;
    c.clear_sphere_radius = config_.clear_sphere_radius;
    c.occupied_sphere_radius = config_.occupied_sphere_radius;
    return c;
  }
  void run(bool batch, bool clear_updated_flag) {
    const vbx_esdf_cfg c = toC();
    const DeviceMap& m = *tsdf_layer_->map();
    m.check(vbx_esdf_update(m.ctx(), &c, batch ? 1 : 0, clear_updated_flag ? 1 : 0), "vbx_esdf_update");
  }
  Config config_;
  Layer<TsdfVoxel>* tsdf_layer_;
  Layer<EsdfVoxel>* esdf_layer_;
};

}  // namespace vbx_host

#endif  // VBX_INTEGRATORS_HPP_
