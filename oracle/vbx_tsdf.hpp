// ORACLE — TEST INFRASTRUCTURE ONLY (see vbx_core.hpp header).
//
// CPU restatement of the voxblox TSDF integrators:
//   include/voxblox/integrator/integrator_utils.h, src/integrator/integrator_utils.cc
//   include/voxblox/utils/approx_hash_array.h
//   include/voxblox/integrator/tsdf_integrator.h, src/integrator/tsdf_integrator.cc
// followed statement by statement, quirks included.  Two switches that are NOT
// in the reference are provided for the GPU parity tests and are off by
// default (see Config::oracle_*): they select deterministic orderings / an
// exact observed-set where the reference's 1-thread behaviour is either
// implementation-defined (libstdc++ unordered_map order) or inherently
// sequential (ApproxHashSet early termination).
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <limits>
#include <list>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_set>

#include "vbx_core.hpp"

namespace orc {

// ---------------------------------------------------------------------------
// ThreadSafeIndex family, integrator_utils.h:22-93, integrator_utils.cc:5-68.
// ---------------------------------------------------------------------------
class ThreadSafeIndex {
 public:
  virtual ~ThreadSafeIndex() = default;
  // integrator_utils.cc:40-50
  bool getNextIndex(size_t* idx) {
    const size_t seq = atomic_idx_.fetch_add(1);
    if (seq >= number_of_points_) return false;
    *idx = getNextIndexImpl(seq);
    return true;
  }

 protected:
  explicit ThreadSafeIndex(size_t n) : atomic_idx_(0), number_of_points_(n) {}
  virtual size_t getNextIndexImpl(size_t seq) = 0;
  std::atomic<size_t> atomic_idx_;
  const size_t number_of_points_;
};

// integrator_utils.cc:20-22, :54-63 — 1024-strided interleave.
inline size_t mixedIndex(size_t seq, size_t number_of_points) {
  constexpr size_t step = 1 << 10;
  const size_t groups = number_of_points / step;
  if (groups * step <= seq) return seq;
  return (seq % groups) * step + seq / groups;
}
class MixedThreadSafeIndex : public ThreadSafeIndex {
 public:
  explicit MixedThreadSafeIndex(size_t n) : ThreadSafeIndex(n) {}

 protected:
  size_t getNextIndexImpl(size_t seq) override {
    return mixedIndex(seq, number_of_points_);
  }
};
// integrator_utils.cc:24-37, :65-67 — ascending squared norm (std::sort, so
// ties are in unspecified order exactly as in the reference).
class SortedThreadSafeIndex : public ThreadSafeIndex {
 public:
  SortedThreadSafeIndex(const Vec3f* pts, size_t n) : ThreadSafeIndex(n) {
    order_.reserve(n);
    for (size_t i = 0; i < n; ++i) order_.emplace_back(i, sqnorm(pts[i]));
    std::sort(order_.begin(), order_.end(),
              [](const std::pair<size_t, double>& a,
                 const std::pair<size_t, double>& b) { return a.second < b.second; });
  }

 protected:
  size_t getNextIndexImpl(size_t seq) override { return order_[seq].first; }
  std::vector<std::pair<size_t, double>> order_;
};

// ---------------------------------------------------------------------------
// RayCaster, integrator_utils.h:101-128, integrator_utils.cc:72-179.
// ---------------------------------------------------------------------------
class RayCaster {
 public:
  // integrator_utils.cc:72-104
  RayCaster(const Vec3f& origin, const Vec3f& point_G, bool is_clearing_ray,
            bool voxel_carving_enabled, float max_ray_length_m,
            float voxel_size_inv, float truncation_distance,
            bool cast_from_origin = true) {
    const Vec3f unit_ray = normalized(point_G - origin);
    Vec3f ray_start, ray_end;
    if (is_clearing_ray) {
      float ray_length = norm(point_G - origin);
      ray_length = std::min(std::max(ray_length - truncation_distance, 0.0f),
                            max_ray_length_m);
      ray_end = origin + unit_ray * ray_length;
      ray_start = voxel_carving_enabled ? origin : ray_end;
    } else {
      ray_end = point_G + unit_ray * truncation_distance;
      ray_start = voxel_carving_enabled ? origin
                                        : (point_G - unit_ray * truncation_distance);
    }
    const Vec3f start_scaled = ray_start * voxel_size_inv;
    const Vec3f end_scaled = ray_end * voxel_size_inv;
    if (cast_from_origin) {
      setup(start_scaled, end_scaled);
    } else {
      setup(end_scaled, start_scaled);
    }
  }
  RayCaster(const Vec3f& start_scaled, const Vec3f& end_scaled) {
    setup(start_scaled, end_scaled);
  }

  // integrator_utils.cc:111-125.  Emits ray_length_in_steps_+1 indices.  The
  // axis choice restates Eigen's minCoeff visitor: strict '<' against the
  // running minimum starting from coefficient 0 (first minimum wins; a NaN in
  // slot 0 is never displaced, a NaN elsewhere never wins).
  bool nextRayIndex(LIdx3* ray_index) {
    if (current_step_++ > ray_length_in_steps_) return false;
    *ray_index = curr_index_;
    int a = 0;
    float m = t_to_next_boundary_[0];
    if (t_to_next_boundary_[1] < m) { m = t_to_next_boundary_[1]; a = 1; }
    if (t_to_next_boundary_[2] < m) { m = t_to_next_boundary_[2]; a = 2; }
    curr_index_[a] += ray_step_signs_[a];
    t_to_next_boundary_[a] += t_step_size_[a];
    return true;
  }
  unsigned ray_length_in_steps() const { return ray_length_in_steps_; }

 private:
  // integrator_utils.cc:127-179.  The `std::abs(x) < 0.0` guards of the
  // reference are never true, so the divisions are unconditional.
  void setup(const Vec3f& start_scaled, const Vec3f& end_scaled) {
    if (std::isnan(start_scaled.x) || std::isnan(start_scaled.y) ||
        std::isnan(start_scaled.z) || std::isnan(end_scaled.x) ||
        std::isnan(end_scaled.y) || std::isnan(end_scaled.z)) {
      // The reference leaves current_step_/curr_index_ uninitialised here
      // (undefined behaviour; the ROS path filters non-finite
      // points first, conversions.h:135-137).  The oracle defines it as
      // "emit nothing", which is also what the HIP path does.
      ray_length_in_steps_ = 0;
      current_step_ = 1;
      return;
    }
    curr_index_ = gridIndexFromScaledPointL(start_scaled);
    const LIdx3 end_index = gridIndexFromScaledPointL(end_scaled);
    const LIdx3 diff{end_index.x - curr_index_.x, end_index.y - curr_index_.y,
                     end_index.z - curr_index_.z};
    current_step_ = 0;
    ray_length_in_steps_ = static_cast<unsigned>(std::abs(diff.x) + std::abs(diff.y) +
                                                 std::abs(diff.z));
    const Vec3f ray_scaled = end_scaled - start_scaled;
    ray_step_signs_ = {signum(ray_scaled.x), signum(ray_scaled.y), signum(ray_scaled.z)};
    const Idx3 corrected{std::max(0, ray_step_signs_.x), std::max(0, ray_step_signs_.y),
                         std::max(0, ray_step_signs_.z)};
    const Vec3f shifted{start_scaled.x - static_cast<float>(curr_index_.x),
                        start_scaled.y - static_cast<float>(curr_index_.y),
                        start_scaled.z - static_cast<float>(curr_index_.z)};
    const Vec3f dist{static_cast<float>(corrected.x) - shifted.x,
                     static_cast<float>(corrected.y) - shifted.y,
                     static_cast<float>(corrected.z) - shifted.z};
    t_to_next_boundary_ = {dist.x / ray_scaled.x, dist.y / ray_scaled.y,
                           dist.z / ray_scaled.z};
    t_step_size_ = {static_cast<float>(ray_step_signs_.x) / ray_scaled.x,
                    static_cast<float>(ray_step_signs_.y) / ray_scaled.y,
                    static_cast<float>(ray_step_signs_.z) / ray_scaled.z};
  }

  Vec3f t_to_next_boundary_{0, 0, 0};
  LIdx3 curr_index_{0, 0, 0};
  Idx3 ray_step_signs_{0, 0, 0};
  Vec3f t_step_size_{0, 0, 0};
  unsigned ray_length_in_steps_ = 0;
  unsigned current_step_ = 0;
};

// ---------------------------------------------------------------------------
// ApproxHashArray / ApproxHashSet, utils/approx_hash_array.h:36-179.
// ---------------------------------------------------------------------------
template <size_t unmasked_bits, typename Stored>
class ApproxHashArray {
 public:
  ApproxHashArray() : map_(size_t(1) << unmasked_bits) {}
  Stored& get(size_t hash) { return map_[hash & ((size_t(1) << unmasked_bits) - 1)]; }

 private:
  std::vector<Stored> map_;
};

template <size_t unmasked_bits, size_t full_reset_threshold>
class ApproxHashSet {
 public:
  // approx_hash_array.h:81-90
  ApproxHashSet() : offset_(0), set_(kSize) {
    for (auto& v : set_) v.store(0, std::memory_order_relaxed);
    set_[offset_].store(std::numeric_limits<size_t>::max());
  }
  // approx_hash_array.h:98-102
  bool isHashCurrentlyPresent(size_t hash) {
    return set_[(hash & kMask) + offset_].load(std::memory_order_relaxed) == hash;
  }
  // approx_hash_array.h:125-134
  bool replaceHash(size_t hash) {
    const size_t i = (hash & kMask) + offset_;
    if (set_[i].load(std::memory_order_relaxed) == hash) return false;
    set_[i].store(hash, std::memory_order_relaxed);
    return true;
  }
  // approx_hash_array.h:156-169
  void resetApproxSet() {
    if (++offset_ >= full_reset_threshold) {
      for (auto& v : set_) v.store(0, std::memory_order_relaxed);
      offset_ = 0;
      set_[offset_].store(std::numeric_limits<size_t>::max());
    }
  }
  size_t offset() const { return offset_; }

 private:
  static constexpr size_t kSize = (size_t(1) << unmasked_bits) + full_reset_threshold;
  static constexpr size_t kMask = (size_t(1) << unmasked_bits) - 1;
  size_t offset_;
  std::vector<std::atomic<size_t>> set_;
};

// ---------------------------------------------------------------------------
// TsdfIntegratorBase, tsdf_integrator.h:51-198, tsdf_integrator.cc:53-240.
// ---------------------------------------------------------------------------
struct TsdfConfig {  // tsdf_integrator.h:56-89
  float default_truncation_distance = 0.1f;
  float max_weight = 10000.0f;
  bool voxel_carving_enabled = true;
  float min_ray_length_m = 0.1f;
  float max_ray_length_m = 5.0f;
  bool use_const_weight = false;
  bool allow_clear = true;
  bool use_weight_dropoff = true;
  bool use_sparsity_compensation_factor = false;
  float sparsity_compensation_factor = 1.0f;
  size_t integrator_threads = std::thread::hardware_concurrency();
  std::string integration_order_mode = "mixed";
  bool enable_anti_grazing = false;
  float start_voxel_subsampling_factor = 2.0f;
  int max_consecutive_ray_collisions = 2;
  int clear_checks_every_n_frames = 1;
  float max_integration_time_s = std::numeric_limits<float>::max();

  // ---- oracle-only switches (NOT in the reference; default = reference) ----
  // Merged: visit bundles in ascending (z,y,x) voxel-key order instead of the
  // libstdc++ unordered_map iteration order (which is implementation-defined).
  bool oracle_merged_sorted_bundles = false;
  // Fast: replace voxel_observed_approx_set_ by an exact (collision-free,
  // eviction-free) set.  The start-voxel set stays the approximate one.
  bool oracle_fast_exact_observed_set = false;
};

struct TsdfStats {  // counters used for the roofline accounting
  std::atomic<uint64_t> voxel_updates{0};   // calls of updateTsdfVoxel
  std::atomic<uint64_t> rays_cast{0};
  void reset() { voxel_updates = 0; rays_cast = 0; }
};

class TsdfIntegratorBase {
 public:
  using TsdfLayer = Layer<TsdfVoxel>;
  using BlockPtr = TsdfLayer::BlockPtr;

  TsdfIntegratorBase(const TsdfConfig& config, TsdfLayer* layer) : config_(config) {
    setLayer(layer);
    // tsdf_integrator.cc:58-65
    if (config_.integrator_threads == 0) config_.integrator_threads = 1;
    if (config_.allow_clear && !config_.voxel_carving_enabled) config_.allow_clear = false;
  }
  virtual ~TsdfIntegratorBase() = default;

  virtual void integratePointCloud(const Transformation& T_G_C, const Vec3f* points_C,
                                   const Color* colors, size_t n,
                                   bool freespace_points = false) = 0;

  // tsdf_integrator.cc:68-80
  void setLayer(TsdfLayer* layer) {
    layer_ = layer;
    voxel_size_ = layer_->voxel_size;
    block_size_ = layer_->block_size;
    voxels_per_side_ = layer_->voxels_per_side;
    voxel_size_inv_ = 1.0 / voxel_size_;
    block_size_inv_ = 1.0 / block_size_;
    voxels_per_side_inv_ = 1.0 / voxels_per_side_;
  }
  const TsdfConfig& getConfig() const { return config_; }
  TsdfStats stats;

 protected:
  // tsdf_integrator.h:112-129
  bool isPointValid(const Vec3f& point_C, bool freespace_point, bool* is_clearing) const {
    const float ray_distance = norm(point_C);
    if (ray_distance < config_.min_ray_length_m) {
      return false;
    } else if (ray_distance > config_.max_ray_length_m) {
      if (config_.allow_clear || freespace_point) {
        *is_clearing = true;
        return true;
      }
      return false;
    } else {
      *is_clearing = freespace_point;
      return true;
    }
  }

  // tsdf_integrator.cc:91-134
  TsdfVoxel* allocateStorageAndGetVoxelPtr(const LIdx3& g, BlockPtr* last_block,
                                           Idx3* last_block_idx) {
    const Idx3 block_idx = blockIndexFromGlobalVoxelIndex(g, voxels_per_side_inv_);
    if ((block_idx != *last_block_idx) || (*last_block == nullptr)) {
      *last_block = layer_->getBlockPtrByIndex(block_idx);
      *last_block_idx = block_idx;
    }
    if (*last_block == nullptr) {
      std::lock_guard<std::mutex> lock(temp_block_mutex_);
      auto it = temp_block_map_.find(block_idx);
      if (it != temp_block_map_.end()) {
        *last_block = it->second;
      } else {
        auto st = temp_block_map_.emplace(
            block_idx, std::make_shared<Block<TsdfVoxel>>(
                           voxels_per_side_, voxel_size_,
                           originPointFromGridIndex(block_idx, block_size_)));
        *last_block = st.first->second;
      }
    }
    (*last_block)->updated = 0x7;  // updated().set() — all three bits (:128)
    const Idx3 local = localFromGlobalVoxelIndex(g, static_cast<int>(voxels_per_side_));
    return &(*last_block)->voxels[(*last_block)->linearIndex(local)];
  }

  // tsdf_integrator.cc:137-147
  void updateLayerWithStoredBlocks() {
    for (const auto& kv : temp_block_map_) layer_->block_map.insert(kv);
    temp_block_map_.clear();
  }

  // tsdf_integrator.cc:216-228
  float computeDistance(const Vec3f& origin, const Vec3f& point_G,
                        const Vec3f& voxel_center) const {
    const Vec3f v_voxel_origin = voxel_center - origin;
    const Vec3f v_point_origin = point_G - origin;
    const float dist_G = norm(v_point_origin);
    const float dist_G_V = dot(v_voxel_origin, v_point_origin) / dist_G;
    return dist_G - dist_G_V;
  }

  // tsdf_integrator.cc:231-240
  float getVoxelWeight(const Vec3f& point_C) const {
    if (config_.use_const_weight) return 1.0f;
    const float dist_z = std::abs(point_C.z);
    if (dist_z > kEpsilon) return 1.0f / (dist_z * dist_z);
    return 0.0f;
  }

  // tsdf_integrator.cc:150-209
  void updateTsdfVoxel(const Vec3f& origin, const Vec3f& point_G, const LIdx3& g,
                       const Color& color, float weight, TsdfVoxel* v) {
    stats.voxel_updates.fetch_add(1, std::memory_order_relaxed);
    const Vec3f voxel_center = centerPointFromGridIndex(g, voxel_size_);
    const float sdf = computeDistance(origin, point_G, voxel_center);

    float updated_weight = weight;
    const float dropoff_epsilon = voxel_size_;
    if (config_.use_weight_dropoff && sdf < -dropoff_epsilon) {
      updated_weight = weight * (config_.default_truncation_distance + sdf) /
                       (config_.default_truncation_distance - dropoff_epsilon);
      updated_weight = std::max(updated_weight, 0.0f);
    }
    if (config_.use_sparsity_compensation_factor) {
      if (std::abs(sdf) < config_.default_truncation_distance)
        updated_weight *= config_.sparsity_compensation_factor;
    }

    std::lock_guard<std::mutex> lock(mutexes_.get(longIndexHash(g)));

    const float new_weight = v->weight + updated_weight;
    if (new_weight < kFloatEpsilon) return;

    const float new_sdf = (sdf * updated_weight + v->distance * v->weight) / new_weight;

    if (std::abs(sdf) < config_.default_truncation_distance)
      v->color = blendTwoColors(v->color, v->weight, color, updated_weight);

    v->distance = (new_sdf > 0.0) ? std::min(config_.default_truncation_distance, new_sdf)
                                  : std::max(-config_.default_truncation_distance, new_sdf);
    v->weight = std::min(config_.max_weight, new_weight);
  }

  ThreadSafeIndex* makeIndexGetter(const Vec3f* pts, size_t n) const {
    // integrator_utils.cc:5-15
    if (config_.integration_order_mode == "sorted") return new SortedThreadSafeIndex(pts, n);
    return new MixedThreadSafeIndex(n);
  }

  TsdfConfig config_;
  TsdfLayer* layer_;
  float voxel_size_;
  size_t voxels_per_side_;
  float block_size_;
  float voxel_size_inv_, voxels_per_side_inv_, block_size_inv_;
  std::mutex temp_block_mutex_;
  TsdfLayer::BlockMap temp_block_map_;
  ApproxHashArray<12, std::mutex> mutexes_;
};

// ---------------------------------------------------------------------------
// SimpleTsdfIntegrator, tsdf_integrator.cc:242-305.
// ---------------------------------------------------------------------------
class SimpleTsdfIntegrator : public TsdfIntegratorBase {
 public:
  using TsdfIntegratorBase::TsdfIntegratorBase;

  void integratePointCloud(const Transformation& T_G_C, const Vec3f* points_C,
                           const Color* colors, size_t n,
                           bool freespace_points = false) override {
    std::unique_ptr<ThreadSafeIndex> index_getter(makeIndexGetter(points_C, n));
    std::list<std::thread> threads;
    for (size_t i = 0; i < config_.integrator_threads; ++i)
      threads.emplace_back(&SimpleTsdfIntegrator::integrateFunction, this, T_G_C, points_C,
                           colors, freespace_points, index_getter.get());
    for (auto& t : threads) t.join();
    updateLayerWithStoredBlocks();
  }

  void integrateFunction(const Transformation& T_G_C, const Vec3f* points_C,
                         const Color* colors, bool freespace_points,
                         ThreadSafeIndex* index_getter) {
    size_t point_idx;
    while (index_getter->getNextIndex(&point_idx)) {
      const Vec3f& point_C = points_C[point_idx];
      const Color& color = colors[point_idx];
      bool is_clearing;
      if (!isPointValid(point_C, freespace_points, &is_clearing)) continue;

      const Vec3f origin = T_G_C.getPosition();
      const Vec3f point_G = T_G_C * point_C;

      RayCaster ray_caster(origin, point_G, is_clearing, config_.voxel_carving_enabled,
                           config_.max_ray_length_m, voxel_size_inv_,
                           config_.default_truncation_distance);
      stats.rays_cast.fetch_add(1, std::memory_order_relaxed);

      BlockPtr block = nullptr;
      Idx3 block_idx{0, 0, 0};
      LIdx3 g;
      while (ray_caster.nextRayIndex(&g)) {
        TsdfVoxel* voxel = allocateStorageAndGetVoxelPtr(g, &block, &block_idx);
        const float weight = getVoxelWeight(point_C);
        updateTsdfVoxel(origin, point_G, g, color, weight, voxel);
      }
    }
  }
};

// ---------------------------------------------------------------------------
// MergedTsdfIntegrator, tsdf_integrator.cc:307-486.
// ---------------------------------------------------------------------------
class MergedTsdfIntegrator : public TsdfIntegratorBase {
 public:
  using TsdfIntegratorBase::TsdfIntegratorBase;
  using VoxelMap = std::unordered_map<LIdx3, std::vector<size_t>, LongIndexHasher>;

  void integratePointCloud(const Transformation& T_G_C, const Vec3f* points_C,
                           const Color* colors, size_t n,
                           bool freespace_points = false) override {
    VoxelMap voxel_map, clear_map;
    std::unique_ptr<ThreadSafeIndex> index_getter(makeIndexGetter(points_C, n));
    bundleRays(T_G_C, points_C, freespace_points, index_getter.get(), &voxel_map, &clear_map);
    last_num_bundles = voxel_map.size();
    last_num_clear_bundles = clear_map.size();
    integrateRays(T_G_C, points_C, colors, config_.enable_anti_grazing, false, voxel_map,
                  clear_map);
    integrateRays(T_G_C, points_C, colors, config_.enable_anti_grazing, true, voxel_map,
                  clear_map);
  }
  size_t last_num_bundles = 0, last_num_clear_bundles = 0;

 protected:
  // tsdf_integrator.cc:340-371
  void bundleRays(const Transformation& T_G_C, const Vec3f* points_C, bool freespace_points,
                  ThreadSafeIndex* index_getter, VoxelMap* voxel_map, VoxelMap* clear_map) {
    size_t point_idx;
    while (index_getter->getNextIndex(&point_idx)) {
      const Vec3f& point_C = points_C[point_idx];
      bool is_clearing;
      if (!isPointValid(point_C, freespace_points, &is_clearing)) continue;
      const Vec3f point_G = T_G_C * point_C;
      const LIdx3 voxel_index = gridIndexFromPointL(point_G, voxel_size_inv_);
      if (is_clearing) {
        (*clear_map)[voxel_index].push_back(point_idx);
      } else {
        (*voxel_map)[voxel_index].push_back(point_idx);
      }
    }
  }

  // tsdf_integrator.cc:373-432
  void integrateVoxel(const Transformation& T_G_C, const Vec3f* points_C, const Color* colors,
                      bool enable_anti_grazing, bool clearing_ray,
                      const std::pair<LIdx3, std::vector<size_t>>& kv,
                      const VoxelMap& voxel_map) {
    if (kv.second.empty()) return;
    const Vec3f origin = T_G_C.getPosition();
    Color merged_color;
    Vec3f merged_point_C{0, 0, 0};
    float merged_weight = 0.0f;

    for (const size_t pt_idx : kv.second) {
      const Vec3f& point_C = points_C[pt_idx];
      const Color& color = colors[pt_idx];
      const float point_weight = getVoxelWeight(point_C);
      if (point_weight < kEpsilon) continue;
      merged_point_C = (merged_point_C * merged_weight + point_C * point_weight) /
                       (merged_weight + point_weight);
      merged_color = blendTwoColors(merged_color, merged_weight, color, point_weight);
      merged_weight += point_weight;
      if (clearing_ray) break;  // only take first point when clearing
    }

    const Vec3f merged_point_G = T_G_C * merged_point_C;
    RayCaster ray_caster(origin, merged_point_G, clearing_ray, config_.voxel_carving_enabled,
                         config_.max_ray_length_m, voxel_size_inv_,
                         config_.default_truncation_distance);
    stats.rays_cast.fetch_add(1, std::memory_order_relaxed);

    LIdx3 g;
    while (ray_caster.nextRayIndex(&g)) {
      if (enable_anti_grazing) {
        if ((clearing_ray || g != kv.first) && voxel_map.find(g) != voxel_map.end()) continue;
      }
      BlockPtr block = nullptr;
      Idx3 block_idx{0, 0, 0};
      TsdfVoxel* voxel = allocateStorageAndGetVoxelPtr(g, &block, &block_idx);
      updateTsdfVoxel(origin, merged_point_G, g, merged_color, merged_weight, voxel);
    }
  }

  // tsdf_integrator.cc:434-457 (+ oracle-only sorted-bundle order).
  void integrateVoxels(const Transformation& T_G_C, const Vec3f* points_C, const Color* colors,
                       bool enable_anti_grazing, bool clearing_ray, const VoxelMap& voxel_map,
                       const VoxelMap& clear_map, size_t thread_idx) {
    const VoxelMap& m = clearing_ray ? clear_map : voxel_map;
    if (config_.oracle_merged_sorted_bundles) {
      std::vector<const VoxelMap::value_type*> order;
      order.reserve(m.size());
      for (const auto& kv : m) order.push_back(&kv);
      std::sort(order.begin(), order.end(),
                [](const VoxelMap::value_type* a, const VoxelMap::value_type* b) {
                  if (a->first.z != b->first.z) return a->first.z < b->first.z;
                  if (a->first.y != b->first.y) return a->first.y < b->first.y;
                  return a->first.x < b->first.x;
                });
      for (size_t i = 0; i < order.size(); ++i)
        if (((i + thread_idx + 1) % config_.integrator_threads) == 0)
          integrateVoxel(T_G_C, points_C, colors, enable_anti_grazing, clearing_ray,
                         std::pair<LIdx3, std::vector<size_t>>(*order[i]), voxel_map);
      return;
    }
    auto it = m.begin();
    const size_t map_size = m.size();
    for (size_t i = 0; i < map_size; ++i) {
      if (((i + thread_idx + 1) % config_.integrator_threads) == 0)
        integrateVoxel(T_G_C, points_C, colors, enable_anti_grazing, clearing_ray,
                       std::pair<LIdx3, std::vector<size_t>>(*it), voxel_map);
      ++it;
    }
  }

  // tsdf_integrator.cc:459-486
  void integrateRays(const Transformation& T_G_C, const Vec3f* points_C, const Color* colors,
                     bool enable_anti_grazing, bool clearing_ray, const VoxelMap& voxel_map,
                     const VoxelMap& clear_map) {
    if (config_.integrator_threads == 1) {
      integrateVoxels(T_G_C, points_C, colors, enable_anti_grazing, clearing_ray, voxel_map,
                      clear_map, 0);
    } else {
      std::list<std::thread> threads;
      for (size_t i = 0; i < config_.integrator_threads; ++i)
        threads.emplace_back(&MergedTsdfIntegrator::integrateVoxels, this, T_G_C, points_C,
                             colors, enable_anti_grazing, clearing_ray, std::cref(voxel_map),
                             std::cref(clear_map), i);
      for (auto& t : threads) t.join();
    }
    updateLayerWithStoredBlocks();
  }
};

// ---------------------------------------------------------------------------
// FastTsdfIntegrator, tsdf_integrator.h:286-341, tsdf_integrator.cc:488-590.
// ---------------------------------------------------------------------------
// tsdf_integrator.cc:564 — function-static counter shared by all instances.
inline int64_t& fastResetCounter() {
  static int64_t reset_counter = 0;
  return reset_counter;
}

class FastTsdfIntegrator : public TsdfIntegratorBase {
 public:
  using TsdfIntegratorBase::TsdfIntegratorBase;

  void integratePointCloud(const Transformation& T_G_C, const Vec3f* points_C,
                           const Color* colors, size_t n,
                           bool freespace_points = false) override {
    integration_start_time_ = std::chrono::steady_clock::now();
    int64_t& reset_counter = fastResetCounter();
    if ((++reset_counter) >= config_.clear_checks_every_n_frames) {
      reset_counter = 0;
      start_voxel_approx_set_.resetApproxSet();
      voxel_observed_approx_set_.resetApproxSet();
      exact_observed_.clear();
    }
    std::unique_ptr<ThreadSafeIndex> index_getter(makeIndexGetter(points_C, n));
    std::list<std::thread> threads;
    for (size_t i = 0; i < config_.integrator_threads; ++i)
      threads.emplace_back(&FastTsdfIntegrator::integrateFunction, this, T_G_C, points_C,
                           colors, freespace_points, index_getter.get());
    for (auto& t : threads) t.join();
    updateLayerWithStoredBlocks();
  }

  // tsdf_integrator.cc:488-553
  void integrateFunction(const Transformation& T_G_C, const Vec3f* points_C,
                         const Color* colors, bool freespace_points,
                         ThreadSafeIndex* index_getter) {
    size_t point_idx;
    while (index_getter->getNextIndex(&point_idx) &&
           (std::chrono::duration_cast<std::chrono::microseconds>(
                std::chrono::steady_clock::now() - integration_start_time_)
                .count() < config_.max_integration_time_s * 1000000)) {
      const Vec3f& point_C = points_C[point_idx];
      const Color& color = colors[point_idx];
      bool is_clearing;
      if (!isPointValid(point_C, freespace_points, &is_clearing)) continue;

      const Vec3f origin = T_G_C.getPosition();
      const Vec3f point_G = T_G_C * point_C;

      LIdx3 g = gridIndexFromPointL(point_G,
                                    config_.start_voxel_subsampling_factor * voxel_size_inv_);
      if (!start_voxel_approx_set_.replaceHash(longIndexHash(g))) continue;

      constexpr bool cast_from_origin = false;
      RayCaster ray_caster(origin, point_G, is_clearing, config_.voxel_carving_enabled,
                           config_.max_ray_length_m, voxel_size_inv_,
                           config_.default_truncation_distance, cast_from_origin);
      stats.rays_cast.fetch_add(1, std::memory_order_relaxed);

      int64_t consecutive_ray_collisions = 0;
      BlockPtr block = nullptr;
      Idx3 block_idx{0, 0, 0};
      while (ray_caster.nextRayIndex(&g)) {
        bool replaced;
        if (config_.oracle_fast_exact_observed_set) {
          std::lock_guard<std::mutex> lock(exact_mutex_);
          replaced = exact_observed_.insert(g).second;
        } else {
          replaced = voxel_observed_approx_set_.replaceHash(longIndexHash(g));
        }
        if (!replaced) {
          ++consecutive_ray_collisions;
        } else {
          consecutive_ray_collisions = 0;
        }
        if (consecutive_ray_collisions > config_.max_consecutive_ray_collisions) break;

        TsdfVoxel* voxel = allocateStorageAndGetVoxelPtr(g, &block, &block_idx);
        const float weight = getVoxelWeight(point_C);
        updateTsdfVoxel(origin, point_G, g, color, weight, voxel);
      }
    }
  }

 private:
  ApproxHashSet<20, 10000> start_voxel_approx_set_;
  ApproxHashSet<20, 10000> voxel_observed_approx_set_;
  std::unordered_set<LIdx3, LongIndexHasher> exact_observed_;  // oracle-only switch
  std::mutex exact_mutex_;
  std::chrono::time_point<std::chrono::steady_clock> integration_start_time_;
};

}  // namespace orc
