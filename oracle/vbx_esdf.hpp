// ORACLE — TEST INFRASTRUCTURE ONLY (see vbx_core.hpp header).
//
// CPU restatement of the voxblox ESDF integrator:
//   include/voxblox/utils/bucket_queue.h
//   include/voxblox/utils/neighbor_tools.h, src/utils/neighbor_tools.cc
//   include/voxblox/integrator/esdf_integrator.h, src/integrator/esdf_integrator.cc
//   include/voxblox/utils/planning_utils_inl.h:14-61 (sphere voxel lists for
//   addNewRobotPosition, esdf_integrator.cc:25-92)
#pragma once

#include <deque>
#include <queue>
#include <unordered_set>

#include "vbx_core.hpp"

namespace orc {

// ---------------------------------------------------------------------------
// BucketQueue, utils/bucket_queue.h:18-99.
// ---------------------------------------------------------------------------
template <typename T>
class BucketQueue {
 public:
  BucketQueue() : last_bucket_index_(0) {}
  // bucket_queue.h:33-39
  void setNumBuckets(int num_buckets, double max_val) {
    max_val_ = max_val;
    num_buckets_ = num_buckets;
    buckets_.clear();
    buckets_.resize(num_buckets_);
    num_elements_ = 0;
  }
  // bucket_queue.h:41-56 — value clamped from above only; |value| bucketed.
  void push(const T& key, double value) {
    if (value > max_val_) value = max_val_;
    int bucket_index =
        static_cast<int>(std::floor(std::abs(value) / max_val_ * (num_buckets_ - 1)));
    if (bucket_index >= num_buckets_) bucket_index = num_buckets_ - 1;
    if (bucket_index < last_bucket_index_) last_bucket_index_ = bucket_index;
    buckets_[bucket_index].push(key);
    num_elements_++;
  }
  // bucket_queue.h:58-70
  void pop() {
    if (empty()) return;
    while (last_bucket_index_ < num_buckets_ && buckets_[last_bucket_index_].empty())
      last_bucket_index_++;
    if (last_bucket_index_ < num_buckets_) {
      buckets_[last_bucket_index_].pop();
      num_elements_--;
    }
  }
  // bucket_queue.h:72-80
  T front() {
    while (last_bucket_index_ < num_buckets_ && buckets_[last_bucket_index_].empty())
      last_bucket_index_++;
    return buckets_[last_bucket_index_].front();
  }
  bool empty() { return num_elements_ == 0; }
  void clear() {
    buckets_.clear();
    buckets_.resize(num_buckets_);
    last_bucket_index_ = 0;
    num_elements_ = 0;
  }

 private:
  int num_buckets_ = 0;
  double max_val_ = 0;
  std::vector<std::queue<T>> buckets_;
  int last_bucket_index_;
  size_t num_elements_ = 0;
};

// ---------------------------------------------------------------------------
// 26-neighbourhood LUTs, neighbor_tools.cc:8-34 (column order is observable).
// ---------------------------------------------------------------------------
struct NeighborhoodLut {
  static constexpr int kN = 26;
  static const int* offsets_x() {
    static const int v[kN] = {-1, 1, 0, 0, 0, 0, -1, -1, 1, 1, 0, 0, 0,
                              0, -1, 1, -1, 1, -1, -1, -1, -1, 1, 1, 1, 1};
    return v;
  }
  static const int* offsets_y() {
    static const int v[kN] = {0, 0, -1, 1, 0, 0, -1, 1, -1, 1, -1, -1, 1,
                              1, 0, 0, 0, 0, -1, -1, 1, 1, -1, -1, 1, 1};
    return v;
  }
  static const int* offsets_z() {
    static const int v[kN] = {0, 0, 0, 0, -1, 1, 0, 0, 0, 0, -1, 1, -1,
                              1, -1, -1, 1, 1, -1, 1, -1, 1, -1, 1, -1, 1};
    return v;
  }
  static float distance(int i) {
    static const float s2 = std::sqrt(2.0), s3 = std::sqrt(3.0);  // :9-10
    return i < 6 ? 1.0f : (i < 18 ? s2 : s3);
  }
  static Idx3 offset(int i) { return {offsets_x()[i], offsets_y()[i], offsets_z()[i]}; }
};

// ---------------------------------------------------------------------------
// EsdfIntegrator, esdf_integrator.h:24-179, esdf_integrator.cc:7-530.
// ---------------------------------------------------------------------------
struct EsdfConfig {  // esdf_integrator.h:29-78
  bool full_euclidean_distance = false;
  float max_distance_m = 2.0f;
  float min_distance_m = 0.2f;
  float default_distance_m = 2.0f;
  float min_diff_m = 0.001f;
  float min_weight = 1e-6f;
  int num_buckets = 20;
  bool multi_queue = false;
  bool add_occupied_crust = false;
  float clear_sphere_radius = 1.5f;
  float occupied_sphere_radius = 5.0f;
  // ---- oracle-only switch (NOT in the reference; default = reference) ----
  // The sign-mismatch assignment of processOpenSet (esdf_integrator.cc:459-488) depends on
  // the pop order of opposite-sign neighbours, i.e. on libstdc++'s unordered_map block
  // order.  With this switch the same candidate is applied whenever it moves the voxel
  // closer to the surface (monotone, hence order-free) — the form the HIP path implements.
  bool oracle_orderfree_sign_mismatch = false;
  // The reference's wavefront only expands voxels it queued, so a voxel that appears next to
  // settled ones can stay under-relaxed until a later update touches it.  With this switch every
  // update ends with one more wavefront seeded from every observed voxel of the layer, i.e. at
  // the fixed point of the reference's own relaxation rule — the result the HIP path computes
  // (its relaxation is a pull over all neighbours).  Same rules, same arithmetic, no queue gaps.
  bool oracle_unrestricted_wavefront = false;
};

struct EsdfStats {
  uint64_t num_lower = 0, num_raise = 0, num_new = 0;
  uint64_t raised = 0;          // processRaiseSet pops
  uint64_t open_pops = 0;       // processOpenSet pops
  uint64_t relaxations = 0;     // successful neighbour updates (N_relaxed, §8(d))
  uint64_t blocks = 0;          // blocks walked by updateFromTsdfBlocks
};

class EsdfIntegrator {
 public:
  using TsdfLayer = Layer<TsdfVoxel>;
  using EsdfLayer = Layer<EsdfVoxel>;

  EsdfIntegrator(const EsdfConfig& config, TsdfLayer* tsdf_layer, EsdfLayer* esdf_layer)
      : config_(config), tsdf_layer_(tsdf_layer), esdf_layer_(esdf_layer) {
    voxels_per_side_ = esdf_layer_->voxels_per_side;
    voxel_size_ = esdf_layer_->voxel_size;
    open_.setNumBuckets(config_.num_buckets, config_.max_distance_m);
  }

  // esdf_integrator.cc:94-102
  void updateFromTsdfLayerBatch() {
    esdf_layer_->block_map.clear();
    std::vector<Idx3> tsdf_blocks;
    tsdf_layer_->getAllAllocatedBlocks(&tsdf_blocks);
    tsdf_blocks.insert(tsdf_blocks.end(), updated_blocks_.begin(), updated_blocks_.end());
    updated_blocks_.clear();
    updateFromTsdfBlocks(tsdf_blocks, false);
  }

  // planning_utils_inl.h:14-48.  The loop variables are floats stepped by 1.0f.
  // HierarchicalIndexMap (common.h:97-99) is an unordered_map keyed with AnyIndexHash; its
  // iteration order decides the ESDF layer's block insertion order and the queue push order,
  // so the same container with the same hash and insertion sequence is used here.
  using SphereList = std::unordered_map<Idx3, std::vector<Idx3>, AnyIndexHasher>;
  void getSphereAroundPoint(const Vec3f& center, float radius, SphereList* out) const {
    const float voxel_size = voxel_size_;
    const float voxel_size_inv = static_cast<float>(1.0 / voxel_size_);
    const int voxels_per_side = static_cast<int>(voxels_per_side_);
    const LIdx3 center_index = gridIndexFromPointL(center, voxel_size_inv);
    const float radius_in_voxels = radius / voxel_size;
    for (float x = -radius_in_voxels; x <= radius_in_voxels; x++) {
      for (float y = -radius_in_voxels; y <= radius_in_voxels; y++) {
        for (float z = -radius_in_voxels; z <= radius_in_voxels; z++) {
          const Vec3f point_voxel_space{x, y, z};
          if (norm(point_voxel_space) <= radius_in_voxels) {
            const LIdx3 g{static_cast<int64_t>(std::floor(x)) + center_index.x,
                          static_cast<int64_t>(std::floor(y)) + center_index.y,
                          static_cast<int64_t>(std::floor(z)) + center_index.z};
            // common.h:245-255
            const float vps_inv = static_cast<float>(1.0 / voxels_per_side);
            const Idx3 block_index = blockIndexFromGlobalVoxelIndex(g, vps_inv);
            const Idx3 voxel_index = localFromGlobalVoxelIndex(g, voxels_per_side);
            (*out)[block_index].push_back(voxel_index);
          }
        }
      }
    }
  }

  // esdf_integrator.cc:25-92
  void addNewRobotPosition(const Vec3f& position) {
    // inner sphere: unknown or hallucinated -> free
    SphereList inner;
    getSphereAroundPoint(position, config_.clear_sphere_radius, &inner);
    for (const auto& kv : inner) esdf_layer_->allocateBlockPtrByIndex(kv.first);  // planning_utils_inl.h:57-60
    for (const auto& kv : inner) {
      const Idx3& b = kv.first;
      auto block_ptr = esdf_layer_->getBlockPtrByIndex(b);
      for (const Idx3& voxel_index : kv.second) {
        if (!block_ptr->isValidVoxelIndex(voxel_index)) continue;
        EsdfVoxel& esdf_voxel = block_ptr->voxels[block_ptr->linearIndex(voxel_index)];
        if (!esdf_voxel.observed || esdf_voxel.hallucinated) {
          if (esdf_voxel.hallucinated) {
            raise_.push(globalVoxelIndexFromBlockAndVoxelIndex(b, voxel_index,
                                                               static_cast<int>(voxels_per_side_)));
          }
          esdf_voxel.distance = config_.default_distance_m;
          esdf_voxel.observed = true;
          esdf_voxel.hallucinated = true;
          esdf_voxel.parent = {0, 0, 0};
          updated_blocks_.insert(b);
        }
      }
    }
    // outer sphere: remaining unknown -> occupied
    SphereList outer;
    getSphereAroundPoint(position, config_.occupied_sphere_radius, &outer);
    for (const auto& kv : outer) esdf_layer_->allocateBlockPtrByIndex(kv.first);
    for (const auto& kv : outer) {
      const Idx3& b = kv.first;
      auto block_ptr = esdf_layer_->getBlockPtrByIndex(b);
      for (const Idx3& voxel_index : kv.second) {
        if (!block_ptr->isValidVoxelIndex(voxel_index)) continue;
        EsdfVoxel& esdf_voxel = block_ptr->voxels[block_ptr->linearIndex(voxel_index)];
        if (!esdf_voxel.observed) {
          esdf_voxel.distance = -config_.default_distance_m;
          esdf_voxel.observed = true;
          esdf_voxel.hallucinated = true;
          esdf_voxel.parent = {0, 0, 0};
          updated_blocks_.insert(b);
        } else if (!esdf_voxel.in_queue) {
          open_.push(globalVoxelIndexFromBlockAndVoxelIndex(b, voxel_index,
                                                            static_cast<int>(voxels_per_side_)),
                     esdf_voxel.distance);
        }
      }
    }
  }

  // esdf_integrator.cc:104-122
  void updateFromTsdfLayer(bool clear_updated_flag) {
    std::vector<Idx3> tsdf_blocks;
    tsdf_layer_->getAllUpdatedBlocks(kEsdf, &tsdf_blocks);
    tsdf_blocks.insert(tsdf_blocks.end(), updated_blocks_.begin(), updated_blocks_.end());
    updated_blocks_.clear();
    updateFromTsdfBlocks(tsdf_blocks, true);
    if (clear_updated_flag) {
      for (const Idx3& b : tsdf_blocks) {
        auto blk = tsdf_layer_->getBlockPtrByIndex(b);
        if (blk) blk->updated &= ~(1u << kEsdf);
      }
    }
  }

  // esdf_integrator.cc:124-302
  void updateFromTsdfBlocks(const std::vector<Idx3>& tsdf_blocks, bool incremental = false) {
    for (const Idx3& block_index : tsdf_blocks) {
      auto tsdf_block = tsdf_layer_->getBlockPtrByIndex(block_index);
      if (!tsdf_block) continue;
      stats.blocks++;
      auto esdf_block = esdf_layer_->allocateBlockPtrByIndex(block_index);
      esdf_block->updated = 0x1;  // set_updated(true) == bitset<3>(1): kMap only (:147)

      const size_t num_voxels = tsdf_block->num_voxels;
      for (size_t lin = 0; lin < num_voxels; ++lin) {
        const TsdfVoxel& tsdf_voxel = tsdf_block->voxels[lin];
        if (tsdf_voxel.weight < config_.min_weight) {
          if (!incremental && config_.add_occupied_crust) {
            EsdfVoxel& e = esdf_block->voxels[lin];
            e.distance = -config_.default_distance_m;
            e.observed = true;
            e.hallucinated = true;
            e.fixed = false;
          }
          continue;
        }
        EsdfVoxel& esdf_voxel = esdf_block->voxels[lin];
        const Idx3 voxel_index = esdf_block->voxelIndexFromLinear(lin);
        const LIdx3 global_index = globalVoxelIndexFromBlockAndVoxelIndex(
            block_index, voxel_index, static_cast<int>(voxels_per_side_));

        const bool tsdf_fixed = isFixed(tsdf_voxel.distance);
        if (!esdf_voxel.observed || esdf_voxel.hallucinated) {
          if (esdf_voxel.hallucinated) raise_.push(global_index);
          if (tsdf_fixed) {
            esdf_voxel.distance = tsdf_voxel.distance;
            esdf_voxel.fixed = true;
            esdf_voxel.in_queue = true;
            open_.push(global_index, esdf_voxel.distance);
          } else {
            esdf_voxel.distance = signum(tsdf_voxel.distance) * config_.default_distance_m;
            esdf_voxel.fixed = false;
            if (incremental) {
              if (updateVoxelFromNeighbors(global_index)) {
                esdf_voxel.in_queue = true;
                open_.push(global_index, esdf_voxel.distance);
              }
            }
          }
          esdf_voxel.parent = {0, 0, 0};
          stats.num_new++;
        } else {
          if (tsdf_fixed || esdf_voxel.fixed) {
            if (!tsdf_fixed) {
              esdf_voxel.distance = signum(tsdf_voxel.distance) * config_.default_distance_m;
              esdf_voxel.parent = {0, 0, 0};
              esdf_voxel.fixed = false;
              raise_.push(global_index);
              esdf_voxel.in_queue = true;
              open_.push(global_index, esdf_voxel.distance);
              stats.num_raise++;
            } else if ((esdf_voxel.distance > 0.0f &&
                        tsdf_voxel.distance + config_.min_diff_m < esdf_voxel.distance) ||
                       (esdf_voxel.distance <= 0.0f &&
                        tsdf_voxel.distance - config_.min_diff_m > esdf_voxel.distance)) {
              esdf_voxel.fixed = tsdf_fixed;
              if (esdf_voxel.fixed) {
                esdf_voxel.distance = tsdf_voxel.distance;
              } else {
                esdf_voxel.distance = signum(tsdf_voxel.distance) * config_.default_distance_m;
              }
              esdf_voxel.parent = {0, 0, 0};
              esdf_voxel.in_queue = true;
              open_.push(global_index, esdf_voxel.distance);
              stats.num_lower++;
            } else if ((esdf_voxel.distance > 0.0f &&
                        tsdf_voxel.distance - config_.min_diff_m > esdf_voxel.distance) ||
                       (esdf_voxel.distance <= 0.0f &&
                        tsdf_voxel.distance + config_.min_diff_m < esdf_voxel.distance)) {
              esdf_voxel.fixed = tsdf_fixed;
              if (esdf_voxel.fixed) {
                esdf_voxel.distance = tsdf_voxel.distance;
              } else {
                esdf_voxel.distance = signum(tsdf_voxel.distance) * config_.default_distance_m;
              }
              esdf_voxel.parent = {0, 0, 0};
              raise_.push(global_index);
              esdf_voxel.in_queue = true;
              open_.push(global_index, esdf_voxel.distance);
              stats.num_raise++;
            }
          } else if (signum(tsdf_voxel.distance) != signum(esdf_voxel.distance)) {
            if (tsdf_voxel.distance < esdf_voxel.distance) {
              esdf_voxel.distance = signum(tsdf_voxel.distance) * config_.default_distance_m;
              esdf_voxel.parent = {0, 0, 0};
              esdf_voxel.in_queue = true;
              open_.push(global_index, esdf_voxel.distance);
              stats.num_lower++;
            } else {
              esdf_voxel.distance = signum(tsdf_voxel.distance) * config_.default_distance_m;
              esdf_voxel.parent = {0, 0, 0};
              raise_.push(global_index);
              stats.num_raise++;
            }
          }
        }
        esdf_voxel.observed = true;
        esdf_voxel.hallucinated = false;
      }
    }
    processRaiseSet();
    processOpenSet();
    if (config_.oracle_unrestricted_wavefront) {
      for (auto& kv : esdf_layer_->block_map) {
        Block<EsdfVoxel>& b = *kv.second;
        for (size_t lin = 0; lin < b.num_voxels; ++lin) {
          EsdfVoxel& v = b.voxels[lin];
          if (!v.observed) continue;
          v.in_queue = true;
          open_.push(globalVoxelIndexFromBlockAndVoxelIndex(kv.first, b.voxelIndexFromLinear(lin),
                                                            static_cast<int>(esdf_layer_->voxels_per_side)),
                     v.distance);
        }
      }
      processOpenSet();
    }
  }

  // esdf_integrator.cc:305-369
  void processRaiseSet() {
    while (!raise_.empty()) {
      const LIdx3 global_index = raise_.front();
      raise_.pop();
      for (int idx = 0; idx < NeighborhoodLut::kN; ++idx) {
        const Idx3 off = NeighborhoodLut::offset(idx);
        const LIdx3 neighbor_index{global_index.x + off.x, global_index.y + off.y,
                                   global_index.z + off.z};
        EsdfVoxel* nv = esdf_layer_->getVoxelPtrByGlobalIndex(neighbor_index);
        if (nv == nullptr) continue;
        if (!nv->observed || nv->fixed) continue;
        const Idx3 direction = off;  // (neighbor_index - global_index).cast<int>()
        bool is_neighbors_parent =
            (nv->parent == Idx3{-direction.x, -direction.y, -direction.z});
        if (config_.full_euclidean_distance) {
          Vec3f d = normalized(Vec3f{static_cast<float>(nv->parent.x),
                                     static_cast<float>(nv->parent.y),
                                     static_cast<float>(nv->parent.z)});
          const Idx3 r{static_cast<int>(std::round(d.x)), static_cast<int>(std::round(d.y)),
                       static_cast<int>(std::round(d.z))};
          is_neighbors_parent = (r == Idx3{-direction.x, -direction.y, -direction.z});
        }
        if (is_neighbors_parent) {
          nv->distance = signum(nv->distance) * config_.default_distance_m;
          nv->parent = {0, 0, 0};
          raise_.push(neighbor_index);
        } else if (!nv->in_queue) {
          open_.push(neighbor_index, nv->distance);
          nv->in_queue = true;
        }
      }
      stats.raised++;
    }
  }

  // esdf_integrator.cc:371-496
  void processOpenSet() {
    while (!open_.empty()) {
      const LIdx3 global_index = open_.front();
      open_.pop();
      stats.open_pops++;
      EsdfVoxel* voxel = esdf_layer_->getVoxelPtrByGlobalIndex(global_index);
      voxel->in_queue = false;
      if (!voxel->observed || voxel->distance >= config_.max_distance_m ||
          voxel->distance <= -config_.max_distance_m)
        continue;

      for (int idx = 0; idx < NeighborhoodLut::kN; ++idx) {
        const Idx3 direction = NeighborhoodLut::offset(idx);
        const LIdx3 neighbor_index{global_index.x + direction.x, global_index.y + direction.y,
                                   global_index.z + direction.z};
        float distance = NeighborhoodLut::distance(idx) * voxel_size_;
        EsdfVoxel* nv = esdf_layer_->getVoxelPtrByGlobalIndex(neighbor_index);
        if (nv == nullptr) continue;
        if (!nv->observed || nv->fixed) continue;

        Idx3 new_parent{-direction.x, -direction.y, -direction.z};
        if (config_.full_euclidean_distance) {
          new_parent = {voxel->parent.x - direction.x, voxel->parent.y - direction.y,
                        voxel->parent.z - direction.z};
          const Vec3f np{static_cast<float>(new_parent.x), static_cast<float>(new_parent.y),
                         static_cast<float>(new_parent.z)};
          const Vec3f vp{static_cast<float>(voxel->parent.x), static_cast<float>(voxel->parent.y),
                         static_cast<float>(voxel->parent.z)};
          distance = voxel_size_ * (norm(np) - norm(vp));
          if (distance < 0.0) continue;
        }

        if (voxel->distance > 0 && nv->distance > 0) {
          if (voxel->distance + distance + config_.min_diff_m < nv->distance) {
            stats.relaxations++;
            nv->distance = voxel->distance + distance;
            nv->parent = new_parent;
            if (config_.multi_queue || !nv->in_queue) {
              open_.push(neighbor_index, nv->distance);
              nv->in_queue = true;
            }
          }
        } else if (voxel->distance <= 0 && nv->distance <= 0) {
          if (voxel->distance - distance - config_.min_diff_m > nv->distance) {
            stats.relaxations++;
            nv->distance = voxel->distance - distance;
            nv->parent = new_parent;
            if (config_.multi_queue || !nv->in_queue) {
              open_.push(neighbor_index, nv->distance);
              nv->in_queue = true;
            }
          }
        } else {
          const float potential_distance = voxel->distance - signum(voxel->distance) * distance;
          if (config_.oracle_orderfree_sign_mismatch) {
            float cand;
            if (static_cast<float>(signum(potential_distance)) == nv->distance) cand = potential_distance;
            else cand = signum(nv->distance) * distance;
            if (std::abs(cand) < std::abs(nv->distance)) {
              stats.relaxations++;
              nv->distance = cand;
              nv->parent = new_parent;
              if (config_.multi_queue || !nv->in_queue) {
                open_.push(neighbor_index, nv->distance);
                nv->in_queue = true;
              }
            }
          } else if (std::abs(potential_distance - nv->distance) > distance) {
            // esdf_integrator.cc:464 compares signum(int) with the float distance.
            if (static_cast<float>(signum(potential_distance)) == nv->distance) {
              stats.relaxations++;
              nv->distance = potential_distance;
              nv->parent = new_parent;
              if (config_.multi_queue || !nv->in_queue) {
                open_.push(neighbor_index, nv->distance);
                nv->in_queue = true;
              }
            } else {
              stats.relaxations++;
              nv->distance = signum(nv->distance) * distance;
              nv->parent = new_parent;
              if (config_.multi_queue || !nv->in_queue) {
                open_.push(neighbor_index, nv->distance);
                nv->in_queue = true;
              }
            }
          }
        }
      }
    }
  }

  // esdf_integrator.cc:498-530 — LUT distance NOT scaled by voxel size (:508).
  bool updateVoxelFromNeighbors(const LIdx3& global_index) {
    EsdfVoxel* voxel = esdf_layer_->getVoxelPtrByGlobalIndex(global_index);
    for (int idx = 0; idx < NeighborhoodLut::kN; ++idx) {
      const Idx3 off = NeighborhoodLut::offset(idx);
      const LIdx3 neighbor_index{global_index.x + off.x, global_index.y + off.y,
                                 global_index.z + off.z};
      const float distance = NeighborhoodLut::distance(idx);
      EsdfVoxel* nv = esdf_layer_->getVoxelPtrByGlobalIndex(neighbor_index);
      if (nv == nullptr) continue;
      if (!nv->observed || nv->distance >= config_.max_distance_m ||
          nv->distance <= -config_.max_distance_m)
        continue;
      if (signum(nv->distance) == signum(voxel->distance)) {
        if (std::abs(nv->distance) < std::abs(voxel->distance)) {
          voxel->distance = nv->distance + signum(voxel->distance) * distance;
          voxel->parent = {-off.x, -off.y, -off.z};
          return true;
        }
      }
    }
    return false;
  }

  bool isFixed(float d) const { return std::abs(d) < config_.min_distance_m; }
  void clear() {
    updated_blocks_.clear();
    open_.clear();
    raise_ = std::queue<LIdx3>();
  }
  EsdfStats stats;

 protected:
  EsdfConfig config_;
  TsdfLayer* tsdf_layer_;
  EsdfLayer* esdf_layer_;
  BucketQueue<LIdx3> open_;
  std::queue<LIdx3> raise_;
  size_t voxels_per_side_;
  float voxel_size_;
  std::unordered_set<Idx3, AnyIndexHasher> updated_blocks_;
};

}  // namespace orc
