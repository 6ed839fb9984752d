// tsdf_integrator_hip.cc — replaces voxblox/src/integrator/tsdf_integrator.cc.
//
// Same classes, same signatures (include/voxblox/integrator/tsdf_integrator.h, unchanged):
//   TsdfIntegratorFactory::create                  tsdf_integrator.cc:8-46
//   TsdfIntegratorBase::TsdfIntegratorBase/setLayer tsdf_integrator.cc:53-80
//   Simple/Merged/FastTsdfIntegrator::integratePointCloud   :242-305, :307-338, :555-590
// The bodies hand the cloud to the MI355X path through the C-ABI (vbx_tsdf_integrate) and then copy
// the blocks the call touched back into the caller's host Layer, so everything that reads the Layer
// between calls (mesher, publishers, save_map, ESDF on the CPU) keeps working.  Failure = glog CHECK
// (abort), the reference's own error convention.
#include "voxblox/integrator/tsdf_integrator.h"

#include <map>
#include <mutex>
#include <sstream>

#include "device_mirror.h"

namespace voxblox {
namespace hip {
namespace {
std::mutex g_mu;
std::map<const void*, DeviceMirror*>& table() {
  static std::map<const void*, DeviceMirror*> t;
  return t;
}
}  // namespace

DeviceMirror& mirrorOf(Layer<TsdfVoxel>* layer) {
  CHECK_NOTNULL(layer);
  std::lock_guard<std::mutex> lock(g_mu);
  DeviceMirror*& m = table()[layer];
  if (m == nullptr) {
    m = new DeviceMirror;
    vbx_map_cfg cfg;
    cfg.voxel_size = layer->voxel_size();
    cfg.voxels_per_side = static_cast<uint32_t>(layer->voxels_per_side());
    cfg.max_blocks = 0;
    m->ctx = vbx_create(&cfg, /*device=*/0);
    CHECK(m->ctx != nullptr) << vbx_last_error(nullptr);
  } else if (layer->getNumberOfAllocatedBlocks() == 0u) {
    // the host layer is the source of truth for existence: an empty one means removeAllBlocks() or a
    // new Layer at a recycled address
    size_t n = 0;
    CHECK_EQ(vbx_num_blocks(m->ctx, VBX_LAYER_TSDF, &n), VBX_OK) << vbx_last_error(m->ctx);
    if (n) CHECK_EQ(vbx_clear(m->ctx, VBX_LAYER_TSDF), VBX_OK) << vbx_last_error(m->ctx);
  }
  return *m;
}

void releaseMirror(const Layer<TsdfVoxel>* layer) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = table().find(layer);
  if (it == table().end()) return;
  vbx_destroy(it->second->ctx);
  delete it->second;
  table().erase(it);
}

void mirrorTsdfToHost(DeviceMirror& dev, Layer<TsdfVoxel>* layer) {
  static_assert(sizeof(TsdfVoxel) == 12, "TsdfVoxel is {float distance; float weight; Color color}");
  size_t n = 0;
  CHECK_EQ(vbx_blocks_updated(dev.ctx, VBX_LAYER_TSDF, VBX_UPDATE_MAP, nullptr, 0, &n), VBX_OK)
      << vbx_last_error(dev.ctx);
  if (n == 0) return;
  dev.idx.resize(3 * n);
  CHECK_EQ(vbx_blocks_updated(dev.ctx, VBX_LAYER_TSDF, VBX_UPDATE_MAP, dev.idx.data(), n, &n), VBX_OK)
      << vbx_last_error(dev.ctx);
  const size_t nv = layer->voxels_per_side() * layer->voxels_per_side() * layer->voxels_per_side();
  dev.tsdf_staging.resize(n * nv);
  dev.bits.resize(n);
  dev.has_data.resize(n);
  CHECK_EQ(vbx_blocks_download(dev.ctx, VBX_LAYER_TSDF, dev.idx.data(), n, dev.tsdf_staging.data(), dev.bits.data(),
                               dev.has_data.data()),
           VBX_OK)
      << vbx_last_error(dev.ctx);
  for (size_t i = 0; i < n; ++i) {
    const BlockIndex bi(dev.idx[3 * i], dev.idx[3 * i + 1], dev.idx[3 * i + 2]);
    Block<TsdfVoxel>::Ptr block = layer->allocateBlockPtrByIndex(bi);
    const TsdfVoxel* src = dev.tsdf_staging.data() + i * nv;
    for (size_t v = 0; v < nv; ++v) block->getVoxelByLinearIndex(v) = src[v];
    // block->updated().set() on every touched block (tsdf_integrator.cc:128); bits a host consumer has
    // cleared since (mesher: kMesh, ESDF: kEsdf) come back only if the device set them again
    block->updated() |= std::bitset<Update::kCount>(dev.bits[i]);
    block->has_data() = dev.has_data[i] != 0;  // the integrators never set it
  }
  // kMap doubles as the mirror's dirty bit on the device; kMesh / kEsdf stay for the device-side
  // mesher / ESDF
  CHECK_EQ(vbx_clear_updated(dev.ctx, VBX_LAYER_TSDF, VBX_UPDATE_MAP), VBX_OK) << vbx_last_error(dev.ctx);
}

namespace {
vbx_tsdf_cfg toC(const TsdfIntegratorBase::Config& c) {
  vbx_tsdf_cfg o;
  vbx_tsdf_cfg_default(&o);
  o.default_truncation_distance = c.default_truncation_distance;
  o.max_weight = c.max_weight;
  o.voxel_carving_enabled = c.voxel_carving_enabled;
  o.min_ray_length_m = c.min_ray_length_m;
  o.max_ray_length_m = c.max_ray_length_m;
  o.use_const_weight = c.use_const_weight;
  o.allow_clear = c.allow_clear;
  o.use_weight_dropoff = c.use_weight_dropoff;
  o.use_sparsity_compensation_factor = c.use_sparsity_compensation_factor;
  o.sparsity_compensation_factor = c.sparsity_compensation_factor;
  o.integrator_threads = static_cast<int32_t>(c.integrator_threads);
  if (c.integration_order_mode == "sorted") {
    o.integration_order_mode = 1;
  } else {
    CHECK(c.integration_order_mode == "mixed") << "Unknown integration order mode: '" << c.integration_order_mode
                                               << "'!";  // integrator_utils.cc:12
    o.integration_order_mode = 0;
  }
  o.enable_anti_grazing = c.enable_anti_grazing;
  o.start_voxel_subsampling_factor = c.start_voxel_subsampling_factor;
  o.max_consecutive_ray_collisions = c.max_consecutive_ray_collisions;
  o.clear_checks_every_n_frames = c.clear_checks_every_n_frames;
  o.max_integration_time_s = c.max_integration_time_s;
  // merged_bundle_order / fast_observed_set stay 0: the reference's own semantics
  return o;
}

void integrateOnDevice(int kind, const TsdfIntegratorBase::Config& config, Layer<TsdfVoxel>* layer,
                       const Transformation& T_G_C, const Pointcloud& points_C, const Colors& colors,
                       bool freespace_points) {
  CHECK_EQ(points_C.size(), colors.size());  // tsdf_integrator.cc:247
  DeviceMirror& dev = mirrorOf(layer);
  const vbx_tsdf_cfg cfg = toC(config);
  const Point pos = T_G_C.getPosition();
  const auto& q = T_G_C.getRotation().toImplementation();  // Eigen::Quaternionf
  const float quat_wxyz[4] = {q.w(), q.x(), q.y(), q.z()};
  // AlignedVector<Eigen::Vector3f> is contiguous with a 12-byte stride; Color is 4 bytes
  static_assert(sizeof(Point) == 12 && sizeof(Color) == 4, "Pointcloud / Colors are handed over zero-copy");
  CHECK_EQ(vbx_tsdf_integrate(dev.ctx, kind, &cfg, pos.data(), quat_wxyz, points_C.empty() ? nullptr : points_C[0].data(),
                              colors.empty() ? nullptr : &colors[0].r, points_C.size(), freespace_points ? 1 : 0),
           VBX_OK)
      << vbx_last_error(dev.ctx);
  mirrorTsdfToHost(dev, layer);
}
}  // namespace
}  // namespace hip

TsdfIntegratorBase::Ptr TsdfIntegratorFactory::create(const std::string& integrator_type_name,
                                                      const TsdfIntegratorBase::Config& config,
                                                      Layer<TsdfVoxel>* layer) {
  CHECK(!integrator_type_name.empty());
  int integrator_type = 1;
  for (const std::string& valid_integrator_type_name : kTsdfIntegratorTypeNames) {
    if (integrator_type_name == valid_integrator_type_name)
      return create(static_cast<TsdfIntegratorType>(integrator_type), config, layer);
    ++integrator_type;
  }
  LOG(FATAL) << "Unknown TSDF integrator type: " << integrator_type_name;
  return TsdfIntegratorBase::Ptr();
}

TsdfIntegratorBase::Ptr TsdfIntegratorFactory::create(const TsdfIntegratorType integrator_type,
                                                      const TsdfIntegratorBase::Config& config,
                                                      Layer<TsdfVoxel>* layer) {
  CHECK_NOTNULL(layer);
  switch (integrator_type) {
    case TsdfIntegratorType::kSimple: return TsdfIntegratorBase::Ptr(new SimpleTsdfIntegrator(config, layer));
    case TsdfIntegratorType::kMerged: return TsdfIntegratorBase::Ptr(new MergedTsdfIntegrator(config, layer));
    case TsdfIntegratorType::kFast: return TsdfIntegratorBase::Ptr(new FastTsdfIntegrator(config, layer));
    default: LOG(FATAL) << "Unknown TSDF integrator type: " << static_cast<int>(integrator_type); break;
  }
  return TsdfIntegratorBase::Ptr();
}

TsdfIntegratorBase::TsdfIntegratorBase(const Config& config, Layer<TsdfVoxel>* layer) : config_(config) {
  setLayer(layer);
  if (config_.integrator_threads == 0) {
    LOG(WARNING) << "Automatic core count failed, defaulting to 1 threads";
    config_.integrator_threads = 1;
  }
  // clearing rays have no utility if voxel_carving is disabled
  if (config_.allow_clear && !config_.voxel_carving_enabled) config_.allow_clear = false;
}

void TsdfIntegratorBase::setLayer(Layer<TsdfVoxel>* layer) {
  CHECK_NOTNULL(layer);
  layer_ = layer;
  voxel_size_ = layer_->voxel_size();
  block_size_ = layer_->block_size();
  voxels_per_side_ = layer_->voxels_per_side();
  voxel_size_inv_ = 1.0 / voxel_size_;
  block_size_inv_ = 1.0 / block_size_;
  voxels_per_side_inv_ = 1.0 / voxels_per_side_;
}

void SimpleTsdfIntegrator::integratePointCloud(const Transformation& T_G_C, const Pointcloud& points_C,
                                               const Colors& colors, const bool freespace_points) {
  hip::integrateOnDevice(VBX_TSDF_SIMPLE, config_, layer_, T_G_C, points_C, colors, freespace_points);
}

void MergedTsdfIntegrator::integratePointCloud(const Transformation& T_G_C, const Pointcloud& points_C,
                                               const Colors& colors, const bool freespace_points) {
  hip::integrateOnDevice(VBX_TSDF_MERGED, config_, layer_, T_G_C, points_C, colors, freespace_points);
}

void FastTsdfIntegrator::integratePointCloud(const Transformation& T_G_C, const Pointcloud& points_C,
                                             const Colors& colors, const bool freespace_points) {
  hip::integrateOnDevice(VBX_TSDF_FAST, config_, layer_, T_G_C, points_C, colors, freespace_points);
}

std::string TsdfIntegratorBase::Config::print() const {
  std::stringstream ss;
  ss << "================== TSDF Integrator Config (HIP drop-in) ====================\n"
     << " - default_truncation_distance: " << default_truncation_distance << "\n"
     << " - max_weight: " << max_weight << "\n"
     << " - voxel_carving_enabled: " << voxel_carving_enabled << "\n"
     << " - min_ray_length_m: " << min_ray_length_m << "\n"
     << " - max_ray_length_m: " << max_ray_length_m << "\n"
     << " - use_const_weight: " << use_const_weight << "\n"
     << " - allow_clear: " << allow_clear << "\n"
     << " - integration_order_mode: " << integration_order_mode << "\n"
     << "==============================================================================\n";
  return ss.str();
}

}  // namespace voxblox
