// esdf_integrator_hip.cc — replaces voxblox/src/integrator/esdf_integrator.cc.
//
// Same class, same signatures (include/voxblox/integrator/esdf_integrator.h, unchanged):
//   EsdfIntegrator::EsdfIntegrator              esdf_integrator.cc:7-22
//   addNewRobotPosition                          :25-92
//   updateFromTsdfLayerBatch / updateFromTsdfLayer / updateFromTsdfBlocks   :94-302
//   processRaiseSet / processOpenSet / updateVoxelFromNeighbors             :305-530 (device wavefront)
// The TSDF the update reads is the device copy the HIP TSDF integrators maintain for the same
// Layer<TsdfVoxel> (device_mirror.h); afterwards the touched ESDF blocks are copied into the host
// Layer<EsdfVoxel>, and the kEsdf bits the reference clears on the host TSDF blocks are cleared there too.
#include "voxblox/integrator/esdf_integrator.h"

#include <cstdlib>
#include <cstring>

#include <atomic>

#include "device_mirror.h"
#include "voxblox/utils/timing.h"

namespace voxblox {
namespace hip {

void mirrorEsdfToHost(DeviceMirror& dev, Layer<EsdfVoxel>* layer) {
  static_assert(sizeof(EsdfVoxel) == 20, "EsdfVoxel is {float distance; bool observed, hallucinated, in_queue, fixed; Vector3i parent}");
  size_t n = 0;
  CHECK_EQ(vbx_blocks_updated(dev.ctx, VBX_LAYER_ESDF, VBX_UPDATE_DIRTY, nullptr, 0, &n), VBX_OK)
      << vbx_last_error(dev.ctx);
  if (n == 0) return;
  dev.idx.resize(3 * n);
  CHECK_EQ(vbx_blocks_updated(dev.ctx, VBX_LAYER_ESDF, VBX_UPDATE_DIRTY, dev.idx.data(), n, &n), VBX_OK)
      << vbx_last_error(dev.ctx);
  const size_t nv = layer->voxels_per_side() * layer->voxels_per_side() * layer->voxels_per_side();
  EsdfVoxel* staging = static_cast<EsdfVoxel*>(dev.down_staging.ensure(n * nv * sizeof(EsdfVoxel)));   // page-locked
  dev.bits.resize(n);
  dev.has_data.resize(n);
  CHECK_EQ(vbx_blocks_download(dev.ctx, VBX_LAYER_ESDF, dev.idx.data(), n, staging, dev.bits.data(), dev.has_data.data()), VBX_OK)
      << vbx_last_error(dev.ctx);
  std::vector<Block<EsdfVoxel>::Ptr> blocks(n);
  std::vector<uint64_t> fps(n), fps_s(n);
  for (size_t i = 0; i < n; ++i)
    blocks[i] = layer->allocateBlockPtrByIndex(BlockIndex(dev.idx[3 * i], dev.idx[3 * i + 1], dev.idx[3 * i + 2]));
  parallelFor(n, [&](size_t i) {
    const EsdfVoxel* src = staging + i * nv;
    std::memcpy(static_cast<void*>(&blocks[i]->getVoxelByLinearIndex(0)), src, nv * sizeof(EsdfVoxel));
    fps[i] = voxelFingerprint(src, nv * sizeof(EsdfVoxel), 0);
    fps_s[i] = voxelFingerprint(src, nv * sizeof(EsdfVoxel), kSampledLines);
  });
  for (size_t i = 0; i < n; ++i) {
    const BlockIndex bi(dev.idx[3 * i], dev.idx[3 * i + 1], dev.idx[3 * i + 2]);
    Block<EsdfVoxel>::Ptr& block = blocks[i];
    block->updated() |= std::bitset<Update::kCount>(dev.bits[i]);  // set_updated(true): kMap only (:147)
    HostBlockRecord<EsdfVoxel>& rec = dev.esdf_known[bi];
    rec.block = block;
    rec.bits = static_cast<uint8_t>(block->updated().to_ulong());
    rec.fingerprint = fps[i];
    rec.sampled = fps_s[i];
  }
  // the device's copy of the block's Update bits is only a carrier towards the host (nothing on the device reads an
  // ESDF block's bits): clear it with the dirty mark, so that a block the wavefront touches later does not bring a
  // stale kMap back
  CHECK_EQ(vbx_clear_updated(dev.ctx, VBX_LAYER_ESDF, VBX_UPDATE_MAP | VBX_UPDATE_DIRTY), VBX_OK) << vbx_last_error(dev.ctx);
}

std::atomic<int>& esdfReferenceOrder() {
  static std::atomic<int> v([] {
    const char* e = getenv("VBX_ESDF_REFERENCE_ORDER");
    return (e && e[0]) ? (e[0] != '0' ? 1 : 0) : 1;
  }());
  return v;
}

namespace {

vbx_esdf_cfg toC(const EsdfIntegrator::Config& c) {
  vbx_esdf_cfg o;
  vbx_esdf_cfg_default(&o);
  o.full_euclidean_distance = c.full_euclidean_distance;
  o.max_distance_m = c.max_distance_m;
  o.min_distance_m = c.min_distance_m;
  o.default_distance_m = c.default_distance_m;
  o.min_diff_m = c.min_diff_m;
  o.min_weight = c.min_weight;
  o.num_buckets = c.num_buckets;
  o.multi_queue = c.multi_queue;
  o.add_occupied_crust = c.add_occupied_crust;
  o.clear_sphere_radius = c.clear_sphere_radius;
  o.occupied_sphere_radius = c.occupied_sphere_radius;
  // EsdfIntegrator::Config is the reference's struct and has no such field: the drop-in computes the reference's own
  // result (vbx_esdf_cfg::reference_order = 1, the queue order replayed on the device) unless the process asks for the
  // order-free fixed point — vbx_dropin_set_esdf_reference_order(0), or VBX_ESDF_REFERENCE_ORDER=0 in its environment
  o.reference_order = esdfReferenceOrder().load() ? 1 : 0;
  return o;
}

// The reference visits the blocks in the iteration order of the CALLER'S containers (Layer::getAllUpdatedBlocks /
// getAllAllocatedBlocks over std::unordered_map, then updated_blocks_; esdf_integrator.cc:96-101, :105-109): in
// reference-order mode that order is taken from the host layer and handed down with the call.
// esdf_integrator.cc:139-145: the walk allocates the ESDF block of every listed TSDF block the layer holds, in list order —
// the host ESDF Layer receives its new blocks in that sequence (its container then iterates like a CPU run's)
void allocateEsdfBlocksInWalkOrder(const BlockIndexList& tsdf_blocks, const Layer<TsdfVoxel>& tsdf_layer, Layer<EsdfVoxel>* esdf_layer) {
  for (const BlockIndex& block_index : tsdf_blocks)
    if (tsdf_layer.hasBlock(block_index)) esdf_layer->allocateBlockPtrByIndex(block_index);
}

std::vector<int32_t> flatten(const BlockIndexList& l) {
  std::vector<int32_t> idx;
  idx.reserve(3 * l.size());
  for (const BlockIndex& b : l) {
    idx.push_back(b.x());
    idx.push_back(b.y());
    idx.push_back(b.z());
  }
  return idx;
}
}  // namespace
}  // namespace hip

EsdfIntegrator::EsdfIntegrator(const Config& config, Layer<TsdfVoxel>* tsdf_layer, Layer<EsdfVoxel>* esdf_layer)
    : config_(config), tsdf_layer_(tsdf_layer), esdf_layer_(esdf_layer) {
  CHECK(tsdf_layer_);
  CHECK(esdf_layer_);
  voxels_per_side_ = esdf_layer_->voxels_per_side();
  voxel_size_ = esdf_layer_->voxel_size();
  CHECK_EQ(esdf_layer_->voxels_per_side(), tsdf_layer_->voxels_per_side());
  CHECK_NEAR(esdf_layer_->voxel_size(), tsdf_layer_->voxel_size(), 1e-6);
  open_.setNumBuckets(config_.num_buckets, config_.max_distance_m);
}

void EsdfIntegrator::addNewRobotPosition(const Point& position) {
  hip::MirrorRef pinned = hip::mirrorOf(tsdf_layer_);
  hip::DeviceMirror& dev = *pinned;
  hip::reconcileTsdfFromHost(dev, tsdf_layer_);
  hip::reconcileEsdfFromHost(dev, esdf_layer_);
  vbx_esdf_cfg cfg = hip::toC(config_);
  if (dev.esdf_pending && raise_.empty())  // clear() since the last addNewRobotPosition (see below)
    CHECK_EQ(vbx_esdf_integrator_clear(dev.ctx), VBX_OK) << vbx_last_error(dev.ctx);
  else if (dev.esdf_pending)
    cfg.reference_order = dev.esdf_pending_ordered ? 1 : 0;  // (the switch moved between two positions: stay with the first)
  CHECK_EQ(vbx_esdf_add_new_robot_position(dev.ctx, &cfg, position.data()), VBX_OK) << vbx_last_error(dev.ctx);
  dev.esdf_pending = true;
  dev.esdf_pending_ordered = cfg.reference_order != 0;
  if (cfg.reference_order) {
    // updated_blocks_ (esdf_integrator.cc:54, :80) is this class's own IndexSet: the blocks go in one by one, in the
    // sequence the reference inserts them, and the set's iteration order is the reference's by construction
    size_t n = 0;
    CHECK_EQ(vbx_esdf_robot_updated_blocks(dev.ctx, 0, nullptr, 0, &n, 0), VBX_OK) << vbx_last_error(dev.ctx);
    std::vector<int32_t> idx(3 * n + 3);
    CHECK_EQ(vbx_esdf_robot_updated_blocks(dev.ctx, 0, idx.data(), n, &n, /*clear=*/1), VBX_OK) << vbx_last_error(dev.ctx);
    for (size_t i = 0; i < n; ++i) updated_blocks_.insert(BlockIndex(idx[3 * i], idx[3 * i + 1], idx[3 * i + 2]));
  }
  // EsdfIntegrator::clear() is inline in the header and only empties the host containers (esdf_integrator.h:138-142); the
  // queues themselves live on the device, so one placeholder entry in the host's raise_ is how the next call tells
  // "robot spheres pending" from "clear() was called in between"
  if (raise_.empty()) raise_.push(GlobalIndex::Zero());
  hip::mirrorEsdfToHost(dev, esdf_layer_);
}

namespace {
// the integrator's containers after an update: empty (esdf_integrator.cc:100, :109; the queues drain)
template <typename Queue>
void dropPlaceholder(Queue* raise) {
  while (!raise->empty()) raise->pop();
}
}  // namespace

void EsdfIntegrator::updateFromTsdfLayerBatch() {
  hip::MirrorRef pinned = hip::mirrorOf(tsdf_layer_);
  hip::DeviceMirror& dev = *pinned;
  hip::reconcileTsdfFromHost(dev, tsdf_layer_);  // a TSDF layer that was LOADED, not integrated (esdf_server, tsdf_to_esdf)
  const vbx_esdf_cfg cfg = hip::toC(config_);
  esdf_layer_->removeAllBlocks();  // esdf_integrator.cc:95
  dev.esdf_known.clear();          // the batch update drops the device's ESDF layer as well
  if (cfg.reference_order) {
    BlockIndexList tsdf_blocks;
    tsdf_layer_->getAllAllocatedBlocks(&tsdf_blocks);  // :96-97, in the host container's order
    tsdf_blocks.insert(tsdf_blocks.end(), updated_blocks_.begin(), updated_blocks_.end());  // :98-99
    const std::vector<int32_t> idx = hip::flatten(tsdf_blocks);
    hip::allocateEsdfBlocksInWalkOrder(tsdf_blocks, *tsdf_layer_, esdf_layer_);
    CHECK_EQ(vbx_clear(dev.ctx, VBX_LAYER_ESDF), VBX_OK) << vbx_last_error(dev.ctx);  // (forgets queued sphere entries too)
    CHECK_EQ(vbx_esdf_update_blocks(dev.ctx, &cfg, idx.empty() ? nullptr : idx.data(), tsdf_blocks.size(), /*incremental=*/0),
             VBX_OK)
        << vbx_last_error(dev.ctx);
  } else {
    CHECK_EQ(vbx_esdf_update(dev.ctx, &cfg, /*batch=*/1, /*clear_updated_flag=*/0), VBX_OK) << vbx_last_error(dev.ctx);
  }
  dev.esdf_pending = false;
  updated_blocks_.clear();
  dropPlaceholder(&raise_);
  hip::mirrorEsdfToHost(dev, esdf_layer_);
}

void EsdfIntegrator::updateFromTsdfLayer(bool clear_updated_flag) {
  hip::MirrorRef pinned = hip::mirrorOf(tsdf_layer_);
  hip::DeviceMirror& dev = *pinned;
  hip::reconcileTsdfFromHost(dev, tsdf_layer_);
  hip::reconcileEsdfFromHost(dev, esdf_layer_);
  vbx_esdf_cfg cfg = hip::toC(config_);
  if (dev.esdf_pending && raise_.empty()) {  // clear() since addNewRobotPosition
    CHECK_EQ(vbx_esdf_integrator_clear(dev.ctx), VBX_OK) << vbx_last_error(dev.ctx);
    dev.esdf_pending = false;
  }
  if (dev.esdf_pending && (cfg.reference_order != 0) != dev.esdf_pending_ordered) {
    // the sphere work was queued in the other form: this one update follows it
    LOG_FIRST_N(WARNING, 1) << "voxblox HIP drop-in: the ESDF order switch moved between addNewRobotPosition and the update; "
                               "this update runs in the form the sphere work was queued in";
    cfg.reference_order = dev.esdf_pending_ordered ? 1 : 0;
  }
  timing::Timer esdf_timer("esdf");  // esdf_integrator.cc:127
  if (cfg.reference_order) {
    BlockIndexList tsdf_blocks;
    tsdf_layer_->getAllUpdatedBlocks(Update::kEsdf, &tsdf_blocks);  // :105-106, in the host container's order
    tsdf_blocks.insert(tsdf_blocks.end(), updated_blocks_.begin(), updated_blocks_.end());  // :107-108
    const std::vector<int32_t> idx = hip::flatten(tsdf_blocks);
    hip::allocateEsdfBlocksInWalkOrder(tsdf_blocks, *tsdf_layer_, esdf_layer_);
    CHECK_EQ(vbx_esdf_update_blocks(dev.ctx, &cfg, idx.empty() ? nullptr : idx.data(), tsdf_blocks.size(), /*incremental=*/1),
             VBX_OK)
        << vbx_last_error(dev.ctx);
    if (clear_updated_flag)  // every block carrying the bit was in the list
      CHECK_EQ(vbx_clear_updated(dev.ctx, VBX_LAYER_TSDF, VBX_UPDATE_ESDF), VBX_OK) << vbx_last_error(dev.ctx);
  } else {
    CHECK_EQ(vbx_esdf_update(dev.ctx, &cfg, /*batch=*/0, clear_updated_flag ? 1 : 0), VBX_OK) << vbx_last_error(dev.ctx);
  }
  dev.esdf_pending = false;
  updated_blocks_.clear();
  dropPlaceholder(&raise_);
  if (clear_updated_flag) {  // esdf_integrator.cc:113-121, on the host copies of the TSDF blocks
    BlockIndexList tsdf_blocks;
    tsdf_layer_->getAllUpdatedBlocks(Update::kEsdf, &tsdf_blocks);
    for (const BlockIndex& block_index : tsdf_blocks)
      if (tsdf_layer_->hasBlock(block_index)) tsdf_layer_->getBlockByIndex(block_index).updated().reset(Update::kEsdf);
  }
  esdf_timer.Stop();
  timing::Timer mirror_timer("hip/esdf_mirror_to_host");
  hip::mirrorEsdfToHost(dev, esdf_layer_);
  mirror_timer.Stop();
}

void EsdfIntegrator::updateFromTsdfBlocks(const BlockIndexList& tsdf_blocks, bool incremental) {
  hip::MirrorRef pinned = hip::mirrorOf(tsdf_layer_);
  hip::DeviceMirror& dev = *pinned;
  hip::reconcileTsdfFromHost(dev, tsdf_layer_);
  hip::reconcileEsdfFromHost(dev, esdf_layer_);
  vbx_esdf_cfg cfg = hip::toC(config_);
  // the same bookkeeping as updateFromTsdfLayer: clear() since addNewRobotPosition (placeholder in raise_ gone), sphere work
  // queued in the other form than the process-wide switch says now
  if (dev.esdf_pending && raise_.empty()) {
    CHECK_EQ(vbx_esdf_integrator_clear(dev.ctx), VBX_OK) << vbx_last_error(dev.ctx);
    dev.esdf_pending = false;
  }
  if (dev.esdf_pending) cfg.reference_order = dev.esdf_pending_ordered ? 1 : 0;
  const std::vector<int32_t> idx = hip::flatten(tsdf_blocks);
  if (cfg.reference_order) hip::allocateEsdfBlocksInWalkOrder(tsdf_blocks, *tsdf_layer_, esdf_layer_);
  CHECK_EQ(vbx_esdf_update_blocks(dev.ctx, &cfg, idx.empty() ? nullptr : idx.data(), tsdf_blocks.size(), incremental ? 1 : 0),
           VBX_OK)
      << vbx_last_error(dev.ctx);
  // the queues drained (esdf_integrator.cc:296-301); updated_blocks_ is the caller's to compose into the list and is left
  // alone, like in the reference
  dev.esdf_pending = false;
  dropPlaceholder(&raise_);
  hip::mirrorEsdfToHost(dev, esdf_layer_);
}

}  // namespace voxblox

extern "C" void vbx_dropin_set_esdf_reference_order(int on) { voxblox::hip::esdfReferenceOrder().store(on ? 1 : 0); }
extern "C" int vbx_dropin_get_esdf_reference_order() { return voxblox::hip::esdfReferenceOrder().load(); }

namespace voxblox {

// The wavefront runs on the device inside the update calls; the queue-level entry points of the class are
// kept for link compatibility and have nothing left to process.
void EsdfIntegrator::processRaiseSet() {}
void EsdfIntegrator::processOpenSet() {}
bool EsdfIntegrator::updateVoxelFromNeighbors(const GlobalIndex&) { return false; }

}  // namespace voxblox
