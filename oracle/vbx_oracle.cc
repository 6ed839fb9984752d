// ORACLE — TEST INFRASTRUCTURE ONLY.  C API glue over the header-only
// restatement; see vbx_oracle.h.
#include "vbx_oracle.h"

#include <cstring>
#include <memory>

#include "vbx_core.hpp"
#include "vbx_esdf.hpp"
#include "vbx_mesh.hpp"
#include "vbx_tsdf.hpp"

using namespace orc;

struct orc_map {
  orc_map(float vs, uint32_t vps) : tsdf(vs, vps), esdf(vs, vps) {}
  Layer<TsdfVoxel> tsdf;
  Layer<EsdfVoxel> esdf;
};
struct orc_tsdf_integrator {
  std::unique_ptr<TsdfIntegratorBase> impl;
  MergedTsdfIntegrator* merged = nullptr;
};
struct orc_esdf_integrator {
  std::unique_ptr<EsdfIntegrator> impl;
};
struct orc_mesh_layer {
  orc_mesh_layer(orc_map* m) : map(m), mesh(m->tsdf.block_size) {}
  orc_map* map;
  MeshLayer mesh;
};
struct ApproxSetIface {
  virtual ~ApproxSetIface() = default;
  virtual bool replace(size_t h) = 0;
  virtual bool present(size_t h) = 0;
  virtual void reset() = 0;
};
template <size_t B, size_t T>
struct ApproxSetImpl : ApproxSetIface {
  ApproxHashSet<B, T> s;
  bool replace(size_t h) override { return s.replaceHash(h); }
  bool present(size_t h) override { return s.isHashCurrentlyPresent(h); }
  void reset() override { s.resetApproxSet(); }
};
struct orc_approx_set {
  std::unique_ptr<ApproxSetIface> s;
};
struct orc_bucket_queue {
  BucketQueue<size_t> q;
};

extern "C" {

void orc_tsdf_cfg_default(orc_tsdf_cfg* c) {
  TsdfConfig d;
  c->default_truncation_distance = d.default_truncation_distance;
  c->max_weight = d.max_weight;
  c->voxel_carving_enabled = d.voxel_carving_enabled;
  c->min_ray_length_m = d.min_ray_length_m;
  c->max_ray_length_m = d.max_ray_length_m;
  c->use_const_weight = d.use_const_weight;
  c->allow_clear = d.allow_clear;
  c->use_weight_dropoff = d.use_weight_dropoff;
  c->use_sparsity_compensation_factor = d.use_sparsity_compensation_factor;
  c->sparsity_compensation_factor = d.sparsity_compensation_factor;
  c->integrator_threads = static_cast<int32_t>(d.integrator_threads);
  c->integration_order_mode = 0;
  c->enable_anti_grazing = d.enable_anti_grazing;
  c->start_voxel_subsampling_factor = d.start_voxel_subsampling_factor;
  c->max_consecutive_ray_collisions = d.max_consecutive_ray_collisions;
  c->clear_checks_every_n_frames = d.clear_checks_every_n_frames;
  c->max_integration_time_s = d.max_integration_time_s;
  c->oracle_merged_sorted_bundles = 0;
  c->oracle_fast_exact_observed_set = 0;
}

void orc_esdf_cfg_default(orc_esdf_cfg* c) {
  EsdfConfig d;
  c->full_euclidean_distance = d.full_euclidean_distance;
  c->max_distance_m = d.max_distance_m;
  c->min_distance_m = d.min_distance_m;
  c->default_distance_m = d.default_distance_m;
  c->min_diff_m = d.min_diff_m;
  c->min_weight = d.min_weight;
  c->num_buckets = d.num_buckets;
  c->multi_queue = d.multi_queue;
  c->add_occupied_crust = d.add_occupied_crust;
  c->clear_sphere_radius = d.clear_sphere_radius;
  c->occupied_sphere_radius = d.occupied_sphere_radius;
  c->oracle_orderfree_sign_mismatch = 0;
  c->oracle_unrestricted_wavefront = 0;
}

static TsdfConfig toCfg(const orc_tsdf_cfg* c) {
  TsdfConfig d;
  d.default_truncation_distance = c->default_truncation_distance;
  d.max_weight = c->max_weight;
  d.voxel_carving_enabled = c->voxel_carving_enabled != 0;
  d.min_ray_length_m = c->min_ray_length_m;
  d.max_ray_length_m = c->max_ray_length_m;
  d.use_const_weight = c->use_const_weight != 0;
  d.allow_clear = c->allow_clear != 0;
  d.use_weight_dropoff = c->use_weight_dropoff != 0;
  d.use_sparsity_compensation_factor = c->use_sparsity_compensation_factor != 0;
  d.sparsity_compensation_factor = c->sparsity_compensation_factor;
  d.integrator_threads = static_cast<size_t>(c->integrator_threads);
  d.integration_order_mode = c->integration_order_mode == 1 ? "sorted" : "mixed";
  d.enable_anti_grazing = c->enable_anti_grazing != 0;
  d.start_voxel_subsampling_factor = c->start_voxel_subsampling_factor;
  d.max_consecutive_ray_collisions = c->max_consecutive_ray_collisions;
  d.clear_checks_every_n_frames = c->clear_checks_every_n_frames;
  d.max_integration_time_s = c->max_integration_time_s;
  d.oracle_merged_sorted_bundles = c->oracle_merged_sorted_bundles != 0;
  d.oracle_fast_exact_observed_set = c->oracle_fast_exact_observed_set != 0;
  return d;
}

orc_map* orc_map_create(float voxel_size, uint32_t vps) { return new orc_map(voxel_size, vps); }
void orc_map_destroy(orc_map* m) { delete m; }

orc_tsdf_integrator* orc_tsdf_integrator_create(orc_map* m, int kind, const orc_tsdf_cfg* cfg) {
  auto* it = new orc_tsdf_integrator;
  const TsdfConfig c = toCfg(cfg);
  switch (kind) {  // TsdfIntegratorFactory::create, tsdf_integrator.cc:26-46
    case 1: it->impl.reset(new SimpleTsdfIntegrator(c, &m->tsdf)); break;
    case 2:
      it->merged = new MergedTsdfIntegrator(c, &m->tsdf);
      it->impl.reset(it->merged);
      break;
    case 3: it->impl.reset(new FastTsdfIntegrator(c, &m->tsdf)); break;
    default: delete it; return nullptr;
  }
  return it;
}
void orc_tsdf_integrator_destroy(orc_tsdf_integrator* it) { delete it; }

int orc_tsdf_integrate(orc_tsdf_integrator* it, const float pos[3], const float q[4],
                       const float* points_C, const uint8_t* rgba, size_t n, int freespace) {
  Transformation T;
  T.t = {pos[0], pos[1], pos[2]};
  T.qw = q[0]; T.qx = q[1]; T.qy = q[2]; T.qz = q[3];
  static_assert(sizeof(Vec3f) == 12 && sizeof(Color) == 4, "layout");
  it->impl->integratePointCloud(T, reinterpret_cast<const Vec3f*>(points_C),
                                reinterpret_cast<const Color*>(rgba), n, freespace != 0);
  return 0;
}

void orc_tsdf_stats(orc_tsdf_integrator* it, uint64_t out[4], int reset) {
  out[0] = it->impl->stats.voxel_updates;
  out[1] = it->impl->stats.rays_cast;
  out[2] = it->merged ? it->merged->last_num_bundles : 0;
  out[3] = it->merged ? it->merged->last_num_clear_bundles : 0;
  if (reset) it->impl->stats.reset();
}
void orc_fast_reset_counter_set(int64_t v) { fastResetCounter() = v; }

orc_mesh_layer* orc_mesh_layer_create(orc_map* m) { return new orc_mesh_layer(m); }
void orc_mesh_layer_destroy(orc_mesh_layer* ml) { delete ml; }
void orc_mesh_generate(orc_mesh_layer* ml, int use_color, float min_weight, int threads, int only_updated,
                       int clear_flag) {
  MeshIntegratorConfig c;
  c.use_color = use_color != 0;
  c.min_weight = min_weight;
  c.integrator_threads = threads > 0 ? threads : 1;
  MeshIntegrator(c, &ml->map->tsdf, &ml->mesh).generateMesh(only_updated != 0, clear_flag != 0);
}
size_t orc_mesh_num_blocks(orc_mesh_layer* ml) { return ml->mesh.mesh_map.size(); }
size_t orc_mesh_block_indices(orc_mesh_layer* ml, int32_t* out, size_t cap) {
  size_t i = 0;
  for (const auto& kv : ml->mesh.mesh_map) {
    if (i < cap) { out[3 * i] = kv.first.x; out[3 * i + 1] = kv.first.y; out[3 * i + 2] = kv.first.z; }
    ++i;
  }
  return i;
}
int orc_mesh_block_sizes(orc_mesh_layer* ml, const int32_t idx[3], uint64_t out[5]) {
  auto m = ml->mesh.getMeshPtrByIndex({idx[0], idx[1], idx[2]});
  if (!m) return 0;
  out[0] = m->vertices.size(); out[1] = m->normals.size(); out[2] = m->colors.size();
  out[3] = m->indices.size(); out[4] = m->updated;
  return 1;
}
int orc_mesh_block_get(orc_mesh_layer* ml, const int32_t idx[3], float* vertices, float* normals, uint8_t* rgba,
                       uint64_t* indices) {
  auto m = ml->mesh.getMeshPtrByIndex({idx[0], idx[1], idx[2]});
  if (!m) return 0;
  static_assert(sizeof(Vec3f) == 12 && sizeof(Color) == 4, "layout");
  if (vertices && !m->vertices.empty()) std::memcpy(vertices, m->vertices.data(), m->vertices.size() * 12);
  if (normals && !m->normals.empty()) std::memcpy(normals, m->normals.data(), m->normals.size() * 12);
  if (rgba && !m->colors.empty()) std::memcpy(rgba, m->colors.data(), m->colors.size() * 4);
  if (indices) for (size_t i = 0; i < m->indices.size(); ++i) indices[i] = m->indices[i];
  return 1;
}
void orc_mesh_clear_updated(orc_mesh_layer* ml) { for (auto& kv : ml->mesh.mesh_map) kv.second->updated = false; }

static EsdfConfig toEsdfCfg(const orc_esdf_cfg* c) {
  EsdfConfig d;
  d.full_euclidean_distance = c->full_euclidean_distance != 0;
  d.max_distance_m = c->max_distance_m;
  d.min_distance_m = c->min_distance_m;
  d.default_distance_m = c->default_distance_m;
  d.min_diff_m = c->min_diff_m;
  d.min_weight = c->min_weight;
  d.num_buckets = c->num_buckets;
  d.multi_queue = c->multi_queue != 0;
  d.add_occupied_crust = c->add_occupied_crust != 0;
  d.clear_sphere_radius = c->clear_sphere_radius;
  d.occupied_sphere_radius = c->occupied_sphere_radius;
  d.oracle_orderfree_sign_mismatch = c->oracle_orderfree_sign_mismatch != 0;
  d.oracle_unrestricted_wavefront = c->oracle_unrestricted_wavefront != 0;
  return d;
}
orc_esdf_integrator* orc_esdf_integrator_create(orc_map* m, const orc_esdf_cfg* cfg) {
  auto* it = new orc_esdf_integrator;
  it->impl.reset(new EsdfIntegrator(toEsdfCfg(cfg), &m->tsdf, &m->esdf));
  return it;
}
void orc_esdf_integrator_destroy(orc_esdf_integrator* it) { delete it; }
void orc_esdf_update_from_tsdf_layer(orc_esdf_integrator* it, int clear_updated_flag) {
  it->impl->updateFromTsdfLayer(clear_updated_flag != 0);
}
void orc_esdf_update_from_tsdf_layer_batch(orc_esdf_integrator* it) {
  it->impl->updateFromTsdfLayerBatch();
}
void orc_esdf_update_from_tsdf_blocks(orc_esdf_integrator* it, const int32_t* idx, size_t n, int incremental) {
  std::vector<Idx3> blocks(n);
  for (size_t i = 0; i < n; ++i) blocks[i] = Idx3{idx[3 * i], idx[3 * i + 1], idx[3 * i + 2]};
  it->impl->updateFromTsdfBlocks(blocks, incremental != 0);
}
void orc_esdf_integrator_clear(orc_esdf_integrator* it) { it->impl->clear(); }
void orc_esdf_add_new_robot_position(orc_esdf_integrator* it, const float p[3]) {
  it->impl->addNewRobotPosition(Vec3f{p[0], p[1], p[2]});
}
void orc_esdf_stats(orc_esdf_integrator* it, uint64_t out[7], int reset) {
  const EsdfStats& s = it->impl->stats;
  out[0] = s.num_lower; out[1] = s.num_raise; out[2] = s.num_new; out[3] = s.raised;
  out[4] = s.open_pops; out[5] = s.relaxations; out[6] = s.blocks;
  if (reset) it->impl->stats = EsdfStats();
}

size_t orc_num_blocks(orc_map* m, int layer) {
  return layer == 0 ? m->tsdf.block_map.size() : m->esdf.block_map.size();
}
size_t orc_block_indices(orc_map* m, int layer, int32_t* out, size_t cap) {
  size_t n = 0;
  auto emit = [&](const Idx3& i) {
    if (n < cap) { out[3 * n] = i.x; out[3 * n + 1] = i.y; out[3 * n + 2] = i.z; }
    ++n;
  };
  if (layer == 0) for (const auto& kv : m->tsdf.block_map) emit(kv.first);
  else for (const auto& kv : m->esdf.block_map) emit(kv.first);
  return n;
}
int orc_tsdf_block_get(orc_map* m, const int32_t idx[3], float* dist, float* weight,
                       uint8_t* rgba, uint8_t* updated_bits) {
  auto b = m->tsdf.getBlockPtrByIndex({idx[0], idx[1], idx[2]});
  if (!b) return 0;
  for (size_t i = 0; i < b->num_voxels; ++i) {
    const TsdfVoxel& v = b->voxels[i];
    if (dist) dist[i] = v.distance;
    if (weight) weight[i] = v.weight;
    if (rgba) { rgba[4 * i] = v.color.r; rgba[4 * i + 1] = v.color.g; rgba[4 * i + 2] = v.color.b; rgba[4 * i + 3] = v.color.a; }
  }
  if (updated_bits) *updated_bits = b->updated;
  return 1;
}
int orc_esdf_block_get(orc_map* m, const int32_t idx[3], float* dist, uint8_t* flags,
                       int32_t* parent, uint8_t* updated_bits) {
  auto b = m->esdf.getBlockPtrByIndex({idx[0], idx[1], idx[2]});
  if (!b) return 0;
  for (size_t i = 0; i < b->num_voxels; ++i) {
    const EsdfVoxel& v = b->voxels[i];
    if (dist) dist[i] = v.distance;
    // flag bits as in Block<EsdfVoxel>::serializeToIntegers, block.cc:204-234
    if (flags) flags[i] = (v.observed ? 1 : 0) | (v.hallucinated ? 2 : 0) | (v.in_queue ? 4 : 0) | (v.fixed ? 8 : 0);
    if (parent) { parent[3 * i] = v.parent.x; parent[3 * i + 1] = v.parent.y; parent[3 * i + 2] = v.parent.z; }
  }
  if (updated_bits) *updated_bits = b->updated;
  return 1;
}
int orc_tsdf_block_set(orc_map* m, const int32_t idx[3], const float* dist, const float* weight,
                       const uint8_t* rgba, uint8_t updated_bits) {
  auto b = m->tsdf.allocateBlockPtrByIndex({idx[0], idx[1], idx[2]});
  for (size_t i = 0; i < b->num_voxels; ++i) {
    TsdfVoxel& v = b->voxels[i];
    v.distance = dist[i];
    v.weight = weight[i];
    v.color.r = rgba[4 * i]; v.color.g = rgba[4 * i + 1]; v.color.b = rgba[4 * i + 2]; v.color.a = rgba[4 * i + 3];
  }
  b->updated = updated_bits;
  return 1;
}
void orc_remove_distant_blocks(orc_map* m, int layer, const float c[3], double max_distance) {
  if (layer == 0) m->tsdf.removeDistantBlocks({c[0], c[1], c[2]}, max_distance);
  else m->esdf.removeDistantBlocks({c[0], c[1], c[2]}, max_distance);
}
void orc_clear(orc_map* m, int layer) {
  if (layer == 0) m->tsdf.block_map.clear(); else m->esdf.block_map.clear();
}
void orc_dropin_stats(orc_map*, uint64_t out[2]) { out[0] = out[1] = 0; }  // no device behind the restatement
uint64_t orc_tsdf_count_observed(orc_map* m) {
  uint64_t n = 0;
  for (const auto& kv : m->tsdf.block_map)
    for (size_t i = 0; i < kv.second->num_voxels; ++i)
      if (kv.second->voxels[i].weight > 1e-6) ++n;
  return n;
}

size_t orc_block_serialize(orc_map* m, int layer, const int32_t idx[3], uint32_t* words, size_t cap) {
  std::vector<uint32_t> data;
  if (layer == 0) {
    auto b = m->tsdf.getBlockPtrByIndex({idx[0], idx[1], idx[2]});
    if (!b) return 0;
    serializeBlock(*b, &data);
  } else {
    auto b = m->esdf.getBlockPtrByIndex({idx[0], idx[1], idx[2]});
    if (!b) return 0;
    serializeBlock(*b, &data);
  }
  for (size_t i = 0; i < data.size() && i < cap; ++i) words[i] = data[i];
  return data.size();
}
int orc_block_deserialize(orc_map* m, int layer, const int32_t idx[3], const uint32_t* words, size_t n) {
  const std::vector<uint32_t> data(words, words + n);
  if (layer == 0) {
    auto b = m->tsdf.allocateBlockPtrByIndex({idx[0], idx[1], idx[2]});
    if (!deserializeBlock(data, b.get())) return 0;
    b->updated = 0x7;
  } else {
    auto b = m->esdf.allocateBlockPtrByIndex({idx[0], idx[1], idx[2]});
    if (!deserializeBlock(data, b.get())) return 0;
    b->updated = 0x7;
  }
  return 1;
}
int orc_esdf_block_set(orc_map* m, const int32_t idx[3], const float* dist, const uint8_t* flags,
                       const int32_t* parent, uint8_t updated_bits) {
  auto b = m->esdf.allocateBlockPtrByIndex({idx[0], idx[1], idx[2]});
  for (size_t i = 0; i < b->num_voxels; ++i) {
    EsdfVoxel& v = b->voxels[i];
    v.distance = dist[i];
    v.observed = flags[i] & 1; v.hallucinated = flags[i] & 2; v.in_queue = flags[i] & 4; v.fixed = flags[i] & 8;
    v.parent = {parent[3 * i], parent[3 * i + 1], parent[3 * i + 2]};
  }
  b->updated = updated_bits;
  return 1;
}

// ---- known-answer helpers ----
void orc_grid_index_from_point(const float p[3], float inv, int64_t out[3]) {
  const LIdx3 r = gridIndexFromPointL({p[0], p[1], p[2]}, inv);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_center_point_from_grid_index(const int64_t idx[3], float gs, float out[3]) {
  const Vec3f r = centerPointFromGridIndex(LIdx3{idx[0], idx[1], idx[2]}, gs);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_origin_point_from_grid_index(const int32_t idx[3], float gs, float out[3]) {
  const Vec3f r = originPointFromGridIndex(Idx3{idx[0], idx[1], idx[2]}, gs);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_grid_index_from_origin_point(const float p[3], float inv, int32_t out[3]) {
  const Idx3 r = gridIndexFromOriginPoint({p[0], p[1], p[2]}, inv);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_block_index_from_global(const int64_t g[3], float vps_inv, int32_t out[3]) {
  const Idx3 r = blockIndexFromGlobalVoxelIndex({g[0], g[1], g[2]}, vps_inv);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_local_from_global(const int64_t g[3], int vps, int32_t out[3]) {
  const Idx3 r = localFromGlobalVoxelIndex({g[0], g[1], g[2]}, vps);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_global_from_block_and_local(const int32_t b[3], const int32_t v[3], int vps, int64_t out[3]) {
  const LIdx3 r = globalVoxelIndexFromBlockAndVoxelIndex({b[0], b[1], b[2]}, {v[0], v[1], v[2]}, vps);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
uint64_t orc_linear_index(const int32_t v[3], int vps) {
  Block<TsdfVoxel> b(vps, 0.1f, {0, 0, 0});
  return b.linearIndex({v[0], v[1], v[2]});
}
void orc_voxel_index_from_linear(uint64_t lin, int vps, int32_t out[3]) {
  Block<TsdfVoxel> b(vps, 0.1f, {0, 0, 0});
  const Idx3 r = b.voxelIndexFromLinear(lin);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
uint64_t orc_any_index_hash(const int32_t i[3]) { return anyIndexHash({i[0], i[1], i[2]}); }
uint64_t orc_long_index_hash(const int64_t i[3]) { return longIndexHash({i[0], i[1], i[2]}); }
uint64_t orc_mixed_index(uint64_t seq, uint64_t n) { return mixedIndex(seq, n); }
uint32_t orc_blend_two_colors(uint32_t a, float w1, uint32_t b, float w2) {
  Color c1, c2;
  std::memcpy(static_cast<void*>(&c1), &a, 4);
  std::memcpy(static_cast<void*>(&c2), &b, 4);
  const Color o = blendTwoColors(c1, w1, c2, w2);
  uint32_t r;
  std::memcpy(&r, &o, 4);
  return r;
}
void orc_transform_point(const float pos[3], const float q[4], const float p[3], float out[3]) {
  Transformation T;
  T.t = {pos[0], pos[1], pos[2]};
  T.qw = q[0]; T.qx = q[1]; T.qy = q[2]; T.qz = q[3];
  const Vec3f r = T * Vec3f{p[0], p[1], p[2]};
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
size_t orc_cast_ray(const float o[3], const float pg[3], int is_clearing, int carving,
                    float max_ray_length_m, float voxel_size_inv, float truncation,
                    int cast_from_origin, int64_t* out, size_t cap) {
  RayCaster rc({o[0], o[1], o[2]}, {pg[0], pg[1], pg[2]}, is_clearing != 0, carving != 0,
               max_ray_length_m, voxel_size_inv, truncation, cast_from_origin != 0);
  size_t n = 0;
  LIdx3 g;
  while (rc.nextRayIndex(&g)) {
    if (n < cap) { out[3 * n] = g.x; out[3 * n + 1] = g.y; out[3 * n + 2] = g.z; }
    ++n;
  }
  return n;
}

orc_approx_set* orc_approx_set_create(int small) {
  auto* r = new orc_approx_set;
  if (small) r->s.reset(new ApproxSetImpl<16, 10>());   // test_approx_hash_array.cc:63
  else r->s.reset(new ApproxSetImpl<20, 10000>());       // tsdf_integrator.h:315-328
  return r;
}
void orc_approx_set_destroy(orc_approx_set* s) { delete s; }
int orc_approx_set_replace_hash(orc_approx_set* s, uint64_t h) { return s->s->replace(h); }
int orc_approx_set_is_present(orc_approx_set* s, uint64_t h) { return s->s->present(h); }
void orc_approx_set_reset(orc_approx_set* s) { s->s->reset(); }

orc_bucket_queue* orc_bucket_queue_create(int nb, double max_val) {
  auto* q = new orc_bucket_queue;
  q->q.setNumBuckets(nb, max_val);
  return q;
}
void orc_bucket_queue_destroy(orc_bucket_queue* q) { delete q; }
void orc_bucket_queue_push(orc_bucket_queue* q, uint64_t key, double value) { q->q.push(key, value); }
uint64_t orc_bucket_queue_front(orc_bucket_queue* q) { return q->q.front(); }
void orc_bucket_queue_pop(orc_bucket_queue* q) { q->q.pop(); }
int orc_bucket_queue_empty(orc_bucket_queue* q) { return q->q.empty(); }

void orc_neighbor_lut(int32_t off[78], float dist[26]) {
  for (int i = 0; i < 26; ++i) {
    const Idx3 o = NeighborhoodLut::offset(i);
    off[3 * i] = o.x; off[3 * i + 1] = o.y; off[3 * i + 2] = o.z;
    dist[i] = NeighborhoodLut::distance(i);
  }
}

}  // extern "C"
