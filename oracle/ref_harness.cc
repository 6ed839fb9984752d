// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// C harness over the REAL reference sources: this file includes voxblox's own headers from
// /root/reference/voxblox/include and is linked with voxblox's own tsdf_integrator.cc,
// integrator_utils.cc, esdf_integrator.cc, neighbor_tools.cc and timing.cc compiled in place
// (see Makefile target `ref`), with Eigen / glog / minkindr / protobuf replaced by the
// minimal stand-ins in ref_shims/.  It exports the same orc_* symbols as liboracle.so for
// the subset the reference can serve, so tests can run identical inputs through both and
// require bit-identical layers (tests/test_oracle_vs_reference_build.py).  What this pins:
// the oracle's restatement of voxblox's code.  What it cannot pin: the arithmetic inside the
// third-party libraries, which both sides restate identically.
#include <cstring>
#include <memory>

#include "voxblox/core/layer.h"
#include "voxblox/integrator/esdf_integrator.h"
#include "voxblox/integrator/tsdf_integrator.h"
#include "voxblox/mesh/mesh_integrator.h"
#include "voxblox/utils/approx_hash_array.h"
#include "voxblox/utils/bucket_queue.h"
#include "voxblox/utils/neighbor_tools.h"

#include "vbx_oracle.h"
#include "voxblox/utils/timing.h"

using namespace voxblox;  // NOLINT

struct orc_map {
  orc_map(float vs, uint32_t vps) : tsdf(vs, vps), esdf(vs, vps) {}
  Layer<TsdfVoxel> tsdf;
  Layer<EsdfVoxel> esdf;
};
struct orc_tsdf_integrator { TsdfIntegratorBase::Ptr impl; };
struct orc_esdf_integrator { std::unique_ptr<EsdfIntegrator> impl; };
struct orc_mesh_layer {
  orc_mesh_layer(orc_map* m) : map(m), mesh(m->tsdf.block_size()) {}
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
  ApproxHashSet<B, T, GlobalIndex, LongIndexHash> s;
  bool replace(size_t h) override { return s.replaceHash(h); }
  bool present(size_t h) override { return s.isHashCurrentlyPresent(h); }
  void reset() override { s.resetApproxSet(); }
};
struct orc_approx_set { std::unique_ptr<ApproxSetIface> s; };
struct orc_bucket_queue { BucketQueue<size_t> q; };

#ifdef VBX_DROPIN
// the mesher's drop-in: MeshIntegrator<TsdfVoxel>::generateMesh specialised for the device (a maintainer adds this include
// at the end of voxblox/mesh/mesh_integrator.h)
#include "mesh_integrator_hip.h"
namespace voxblox { namespace hip {
void releaseMirror(const Layer<TsdfVoxel>* tsdf_layer);
void mirrorStats(const Layer<TsdfVoxel>* tsdf_layer, uint64_t* uploaded_blocks, uint64_t* removed_blocks);
void markLayerEdited(const Layer<TsdfVoxel>* tsdf_layer);
void markLayerEdited(const Layer<EsdfVoxel>* esdf_layer);
} }
#endif

extern "C" {

void orc_tsdf_cfg_default(orc_tsdf_cfg* c) {
  TsdfIntegratorBase::Config d;
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
  c->integration_order_mode = d.integration_order_mode == "sorted" ? 1 : 0;
  c->enable_anti_grazing = d.enable_anti_grazing;
  c->start_voxel_subsampling_factor = d.start_voxel_subsampling_factor;
  c->max_consecutive_ray_collisions = d.max_consecutive_ray_collisions;
  c->clear_checks_every_n_frames = d.clear_checks_every_n_frames;
  c->max_integration_time_s = d.max_integration_time_s;
  c->oracle_merged_sorted_bundles = 0;
  c->oracle_fast_exact_observed_set = 0;
}
void orc_esdf_cfg_default(orc_esdf_cfg* c) {
  EsdfIntegrator::Config d;
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

orc_map* orc_map_create(float voxel_size, uint32_t vps) { return new orc_map(voxel_size, vps); }
#ifdef VBX_DROPIN  // libvbxref_hip.so: the integrators are the HIP drop-in; drop the layer's device map with it
void orc_map_destroy(orc_map* m) {
  voxblox::hip::releaseMirror(&m->tsdf);
  delete m;
}
#else
void orc_map_destroy(orc_map* m) { delete m; }
#endif

orc_tsdf_integrator* orc_tsdf_integrator_create(orc_map* m, int kind, const orc_tsdf_cfg* c) {
  if (c->oracle_merged_sorted_bundles || c->oracle_fast_exact_observed_set) return nullptr;  // reference only
  TsdfIntegratorBase::Config d;
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
  if (kind < 1 || kind > 3) return nullptr;
  auto* it = new orc_tsdf_integrator;
  it->impl = TsdfIntegratorFactory::create(static_cast<TsdfIntegratorType>(kind), d, &m->tsdf);
  return it;
}
void orc_tsdf_integrator_destroy(orc_tsdf_integrator* it) { delete it; }

int orc_tsdf_integrate(orc_tsdf_integrator* it, const float pos[3], const float q[4],
                       const float* points_C, const uint8_t* rgba, size_t n, int freespace) {
  const Transformation T(Rotation(q[0], q[1], q[2], q[3]), Point(pos[0], pos[1], pos[2]));
  Pointcloud pts(n);
  Colors cols(n);
  for (size_t i = 0; i < n; ++i) {
    pts[i] = Point(points_C[3 * i], points_C[3 * i + 1], points_C[3 * i + 2]);
    cols[i] = Color(rgba[4 * i], rgba[4 * i + 1], rgba[4 * i + 2], rgba[4 * i + 3]);
  }
  it->impl->integratePointCloud(T, pts, cols, freespace != 0);
  return 0;
}
void orc_tsdf_stats(orc_tsdf_integrator*, uint64_t out[4], int) { out[0] = out[1] = out[2] = out[3] = 0; }
void orc_fast_reset_counter_set(int64_t) {}  // function-static in the reference; moot for n_frames == 1

orc_esdf_integrator* orc_esdf_integrator_create(orc_map* m, const orc_esdf_cfg* c) {
  if (c->oracle_orderfree_sign_mismatch || c->oracle_unrestricted_wavefront) return nullptr;  // reference only
  EsdfIntegrator::Config d;
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
  auto* it = new orc_esdf_integrator;
  it->impl.reset(new EsdfIntegrator(d, &m->tsdf, &m->esdf));
  return it;
}
void orc_esdf_integrator_destroy(orc_esdf_integrator* it) { delete it; }
void orc_esdf_update_from_tsdf_layer(orc_esdf_integrator* it, int clear) { it->impl->updateFromTsdfLayer(clear != 0); }
void orc_esdf_update_from_tsdf_layer_batch(orc_esdf_integrator* it) { it->impl->updateFromTsdfLayerBatch(); }
void orc_esdf_update_from_tsdf_blocks(orc_esdf_integrator* it, const int32_t* idx, size_t n, int incremental) {
  BlockIndexList blocks;
  for (size_t i = 0; i < n; ++i) blocks.push_back(BlockIndex(idx[3 * i], idx[3 * i + 1], idx[3 * i + 2]));
  it->impl->updateFromTsdfBlocks(blocks, incremental != 0);
}
void orc_esdf_integrator_clear(orc_esdf_integrator* it) { it->impl->clear(); }
void orc_esdf_add_new_robot_position(orc_esdf_integrator* it, const float p[3]) {
  it->impl->addNewRobotPosition(Point(p[0], p[1], p[2]));
}
void orc_esdf_stats(orc_esdf_integrator*, uint64_t out[7], int) { for (int i = 0; i < 7; ++i) out[i] = 0; }

static Mesh::Ptr find_mesh(orc_mesh_layer* ml, const BlockIndex& bi) {
  BlockIndexList l;
  ml->mesh.getAllAllocatedMeshes(&l);
  for (const BlockIndex& b : l) if (b == bi) return ml->mesh.getMeshPtrByIndex(bi);
  return Mesh::Ptr();
}
orc_mesh_layer* orc_mesh_layer_create(orc_map* m) { return new orc_mesh_layer(m); }
void orc_mesh_layer_destroy(orc_mesh_layer* ml) { delete ml; }
void orc_mesh_generate(orc_mesh_layer* ml, int use_color, float min_weight, int threads, int only_updated,
                       int clear_flag) {
  MeshIntegratorConfig c;
  c.use_color = use_color != 0;
  c.min_weight = min_weight;
  c.integrator_threads = threads > 0 ? threads : 1;
  MeshIntegrator<TsdfVoxel> integrator(c, &ml->map->tsdf, &ml->mesh);
  integrator.generateMesh(only_updated != 0, clear_flag != 0);
}
size_t orc_mesh_num_blocks(orc_mesh_layer* ml) { return ml->mesh.getNumberOfAllocatedMeshes(); }
size_t orc_mesh_block_indices(orc_mesh_layer* ml, int32_t* out, size_t cap) {
  BlockIndexList l;
  ml->mesh.getAllAllocatedMeshes(&l);
  for (size_t i = 0; i < l.size() && i < cap; ++i) { out[3 * i] = l[i].x(); out[3 * i + 1] = l[i].y(); out[3 * i + 2] = l[i].z(); }
  return l.size();
}
int orc_mesh_block_sizes(orc_mesh_layer* ml, const int32_t idx[3], uint64_t out[5]) {
  const BlockIndex bi(idx[0], idx[1], idx[2]);
  Mesh::Ptr m = find_mesh(ml, bi);
  if (!m) return 0;
  out[0] = m->vertices.size(); out[1] = m->normals.size(); out[2] = m->colors.size();
  out[3] = m->indices.size(); out[4] = m->updated;
  return 1;
}
int orc_mesh_block_get(orc_mesh_layer* ml, const int32_t idx[3], float* vertices, float* normals, uint8_t* rgba,
                       uint64_t* indices) {
  const BlockIndex bi(idx[0], idx[1], idx[2]);
  Mesh::Ptr m = find_mesh(ml, bi);
  if (!m) return 0;
  for (size_t i = 0; i < m->vertices.size(); ++i)
    for (int k = 0; k < 3; ++k) if (vertices) vertices[3 * i + k] = m->vertices[i][k];
  for (size_t i = 0; i < m->normals.size(); ++i)
    for (int k = 0; k < 3; ++k) if (normals) normals[3 * i + k] = m->normals[i][k];
  for (size_t i = 0; i < m->colors.size(); ++i) if (rgba) {
    rgba[4 * i] = m->colors[i].r; rgba[4 * i + 1] = m->colors[i].g; rgba[4 * i + 2] = m->colors[i].b; rgba[4 * i + 3] = m->colors[i].a;
  }
  if (indices) for (size_t i = 0; i < m->indices.size(); ++i) indices[i] = m->indices[i];
  return 1;
}
void orc_mesh_clear_updated(orc_mesh_layer* ml) {
  BlockIndexList l;
  ml->mesh.getAllAllocatedMeshes(&l);
  for (const BlockIndex& b : l) ml->mesh.getMeshPtrByIndex(b)->updated = false;
}

size_t orc_num_blocks(orc_map* m, int layer) {
  return layer == 0 ? m->tsdf.getNumberOfAllocatedBlocks() : m->esdf.getNumberOfAllocatedBlocks();
}
size_t orc_block_indices(orc_map* m, int layer, int32_t* out, size_t cap) {
  BlockIndexList l;
  if (layer == 0) m->tsdf.getAllAllocatedBlocks(&l); else m->esdf.getAllAllocatedBlocks(&l);
  for (size_t i = 0; i < l.size() && i < cap; ++i) { out[3 * i] = l[i].x(); out[3 * i + 1] = l[i].y(); out[3 * i + 2] = l[i].z(); }
  return l.size();
}
int orc_tsdf_block_get(orc_map* m, const int32_t idx[3], float* dist, float* weight, uint8_t* rgba,
                       uint8_t* updated_bits) {
  Block<TsdfVoxel>::Ptr b = m->tsdf.getBlockPtrByIndex(BlockIndex(idx[0], idx[1], idx[2]));
  if (!b) return 0;
  for (size_t i = 0; i < b->num_voxels(); ++i) {
    const TsdfVoxel& v = b->getVoxelByLinearIndex(i);
    if (dist) dist[i] = v.distance;
    if (weight) weight[i] = v.weight;
    if (rgba) { rgba[4 * i] = v.color.r; rgba[4 * i + 1] = v.color.g; rgba[4 * i + 2] = v.color.b; rgba[4 * i + 3] = v.color.a; }
  }
  if (updated_bits) *updated_bits = static_cast<uint8_t>(b->updated().to_ulong());
  return 1;
}
int orc_esdf_block_get(orc_map* m, const int32_t idx[3], float* dist, uint8_t* flags, int32_t* parent,
                       uint8_t* updated_bits) {
  Block<EsdfVoxel>::Ptr b = m->esdf.getBlockPtrByIndex(BlockIndex(idx[0], idx[1], idx[2]));
  if (!b) return 0;
  for (size_t i = 0; i < b->num_voxels(); ++i) {
    const EsdfVoxel& v = b->getVoxelByLinearIndex(i);
    if (dist) dist[i] = v.distance;
    if (flags) flags[i] = (v.observed ? 1 : 0) | (v.hallucinated ? 2 : 0) | (v.in_queue ? 4 : 0) | (v.fixed ? 8 : 0);
    if (parent) { parent[3 * i] = v.parent.x(); parent[3 * i + 1] = v.parent.y(); parent[3 * i + 2] = v.parent.z(); }
  }
  if (updated_bits) *updated_bits = static_cast<uint8_t>(b->updated().to_ulong());
  return 1;
}
int orc_tsdf_block_set(orc_map* m, const int32_t idx[3], const float* dist, const float* weight,
                       const uint8_t* rgba, uint8_t updated_bits) {
  Block<TsdfVoxel>::Ptr b = m->tsdf.allocateBlockPtrByIndex(BlockIndex(idx[0], idx[1], idx[2]));
  for (size_t i = 0; i < b->num_voxels(); ++i) {
    TsdfVoxel& v = b->getVoxelByLinearIndex(i);
    v.distance = dist[i]; v.weight = weight[i];
    v.color = Color(rgba[4 * i], rgba[4 * i + 1], rgba[4 * i + 2], rgba[4 * i + 3]);
  }
  b->updated() = std::bitset<Update::kCount>(updated_bits);
  return 1;
}
void orc_remove_distant_blocks(orc_map* m, int layer, const float c[3], double max_distance) {
  if (layer == 0) m->tsdf.removeDistantBlocks(Point(c[0], c[1], c[2]), max_distance);
  else m->esdf.removeDistantBlocks(Point(c[0], c[1], c[2]), max_distance);
}
void orc_clear(orc_map* m, int layer) { if (layer == 0) m->tsdf.removeAllBlocks(); else m->esdf.removeAllBlocks(); }
// voxblox::timing (utils/timing.h:132-194): samples and total seconds of a tag; count 0 when the tag does not exist
void orc_timing_get(const char* tag, double out[2]) {
  out[0] = out[1] = 0.0;
  try {
    out[0] = (double)voxblox::timing::Timing::GetNumSamples(std::string(tag));
    out[1] = voxblox::timing::Timing::GetTotalSeconds(std::string(tag));
  } catch (...) {
  }
}
void orc_timing_reset() { voxblox::timing::Timing::Reset(); }
void orc_dropin_stats(orc_map* m, uint64_t out[2]) {
  out[0] = out[1] = 0;
#ifdef VBX_DROPIN
  voxblox::hip::mirrorStats(&m->tsdf, &out[0], &out[1]);
#else
  (void)m;
#endif
}
// hip::markLayerEdited: the caller (a test) wrote voxels of the layer in place without touching Update bits
void orc_dropin_mark_edited(orc_map* m, int layer) {
#ifdef VBX_DROPIN
  if (layer == 0) voxblox::hip::markLayerEdited(&m->tsdf); else voxblox::hip::markLayerEdited(&m->esdf);
#else
  (void)m; (void)layer;
#endif
}
// 1: MeshIntegrator<TsdfVoxel>::generateMesh of this library is the HIP specialisation (mesh_integrator_hip.h)
int orc_dropin_mesher() {
#ifdef VBX_DROPIN
  return 1;
#else
  return 0;
#endif
}
uint64_t orc_tsdf_count_observed(orc_map* m) {
  uint64_t n = 0;
  BlockIndexList l;
  m->tsdf.getAllAllocatedBlocks(&l);
  for (const BlockIndex& bi : l) {
    const Block<TsdfVoxel>& b = m->tsdf.getBlockByIndex(bi);
    for (size_t i = 0; i < b.num_voxels(); ++i) if (b.getVoxelByLinearIndex(i).weight > 1e-6) ++n;
  }
  return n;
}

size_t orc_block_serialize(orc_map* m, int layer, const int32_t idx[3], uint32_t* words, size_t cap) {
  std::vector<uint32_t> data;
  const BlockIndex bi(idx[0], idx[1], idx[2]);
  if (layer == 0) {
    Block<TsdfVoxel>::Ptr b = m->tsdf.getBlockPtrByIndex(bi);
    if (!b) return 0;
    b->serializeToIntegers(&data);
  } else {
    Block<EsdfVoxel>::Ptr b = m->esdf.getBlockPtrByIndex(bi);
    if (!b) return 0;
    b->serializeToIntegers(&data);
  }
  for (size_t i = 0; i < data.size() && i < cap; ++i) words[i] = data[i];
  return data.size();
}
int orc_block_deserialize(orc_map* m, int layer, const int32_t idx[3], const uint32_t* words, size_t n) {
  const std::vector<uint32_t> data(words, words + n);
  const BlockIndex bi(idx[0], idx[1], idx[2]);
  if (layer == 0) {
    Block<TsdfVoxel>::Ptr b = m->tsdf.allocateBlockPtrByIndex(bi);
    if (data.size() != b->num_voxels() * 3) return 0;
    b->deserializeFromIntegers(data);
    b->updated().set();
  } else {
    Block<EsdfVoxel>::Ptr b = m->esdf.allocateBlockPtrByIndex(bi);
    if (data.size() != b->num_voxels() * 2) return 0;
    b->deserializeFromIntegers(data);
    b->updated().set();
  }
  return 1;
}
int orc_esdf_block_set(orc_map* m, const int32_t idx[3], const float* dist, const uint8_t* flags,
                       const int32_t* parent, uint8_t updated_bits) {
  Block<EsdfVoxel>::Ptr b = m->esdf.allocateBlockPtrByIndex(BlockIndex(idx[0], idx[1], idx[2]));
  for (size_t i = 0; i < b->num_voxels(); ++i) {
    EsdfVoxel& v = b->getVoxelByLinearIndex(i);
    v.distance = dist[i];
    v.observed = flags[i] & 1; v.hallucinated = flags[i] & 2; v.in_queue = flags[i] & 4; v.fixed = flags[i] & 8;
    v.parent = Eigen::Vector3i(parent[3 * i], parent[3 * i + 1], parent[3 * i + 2]);
  }
  b->updated() = std::bitset<Update::kCount>(updated_bits);
  return 1;
}

void orc_grid_index_from_point(const float p[3], float inv, int64_t out[3]) {
  const GlobalIndex r = getGridIndexFromPoint<GlobalIndex>(Point(p[0], p[1], p[2]), inv);
  out[0] = r.x(); out[1] = r.y(); out[2] = r.z();
}
void orc_center_point_from_grid_index(const int64_t idx[3], float gs, float out[3]) {
  const Point r = getCenterPointFromGridIndex(GlobalIndex(idx[0], idx[1], idx[2]), gs);
  out[0] = r.x(); out[1] = r.y(); out[2] = r.z();
}
void orc_origin_point_from_grid_index(const int32_t idx[3], float gs, float out[3]) {
  const Point r = getOriginPointFromGridIndex(BlockIndex(idx[0], idx[1], idx[2]), gs);
  out[0] = r.x(); out[1] = r.y(); out[2] = r.z();
}
void orc_grid_index_from_origin_point(const float p[3], float inv, int32_t out[3]) {
  const BlockIndex r = getGridIndexFromOriginPoint<BlockIndex>(Point(p[0], p[1], p[2]), inv);
  out[0] = r.x(); out[1] = r.y(); out[2] = r.z();
}
void orc_block_index_from_global(const int64_t g[3], float vps_inv, int32_t out[3]) {
  const BlockIndex r = getBlockIndexFromGlobalVoxelIndex(GlobalIndex(g[0], g[1], g[2]), vps_inv);
  out[0] = r.x(); out[1] = r.y(); out[2] = r.z();
}
void orc_local_from_global(const int64_t g[3], int vps, int32_t out[3]) {
  const VoxelIndex r = getLocalFromGlobalVoxelIndex(GlobalIndex(g[0], g[1], g[2]), vps);
  out[0] = r.x(); out[1] = r.y(); out[2] = r.z();
}
void orc_global_from_block_and_local(const int32_t b[3], const int32_t v[3], int vps, int64_t out[3]) {
  const GlobalIndex r = getGlobalVoxelIndexFromBlockAndVoxelIndex(BlockIndex(b[0], b[1], b[2]), VoxelIndex(v[0], v[1], v[2]), vps);
  out[0] = r.x(); out[1] = r.y(); out[2] = r.z();
}
uint64_t orc_linear_index(const int32_t v[3], int vps) {
  Block<TsdfVoxel> b(vps, 0.1f, Point(0, 0, 0));
  return b.computeLinearIndexFromVoxelIndex(VoxelIndex(v[0], v[1], v[2]));
}
void orc_voxel_index_from_linear(uint64_t lin, int vps, int32_t out[3]) {
  Block<TsdfVoxel> b(vps, 0.1f, Point(0, 0, 0));
  const VoxelIndex r = b.computeVoxelIndexFromLinearIndex(lin);
  out[0] = r.x(); out[1] = r.y(); out[2] = r.z();
}
uint64_t orc_any_index_hash(const int32_t i[3]) { return AnyIndexHash()(AnyIndex(i[0], i[1], i[2])); }
uint64_t orc_long_index_hash(const int64_t i[3]) { return LongIndexHash()(LongIndex(i[0], i[1], i[2])); }
uint64_t orc_mixed_index(uint64_t seq, uint64_t n) {
  // MixedThreadSafeIndex hands out indices through an atomic counter: pull seq+1 of them.
  MixedThreadSafeIndex idx(n);
  size_t out = 0;
  for (uint64_t i = 0; i <= seq; ++i) idx.getNextIndex(&out);
  return out;
}
uint32_t orc_blend_two_colors(uint32_t a, float w1, uint32_t b, float w2) {
  const Color c1(a & 0xFF, (a >> 8) & 0xFF, (a >> 16) & 0xFF, (a >> 24) & 0xFF);
  const Color c2(b & 0xFF, (b >> 8) & 0xFF, (b >> 16) & 0xFF, (b >> 24) & 0xFF);
  const Color o = Color::blendTwoColors(c1, w1, c2, w2);
  return o.r | (o.g << 8) | (o.b << 16) | (static_cast<uint32_t>(o.a) << 24);
}
void orc_transform_point(const float pos[3], const float q[4], const float p[3], float out[3]) {
  const Transformation T(Rotation(q[0], q[1], q[2], q[3]), Point(pos[0], pos[1], pos[2]));
  const Point r = T * Point(p[0], p[1], p[2]);
  out[0] = r.x(); out[1] = r.y(); out[2] = r.z();
}
size_t orc_cast_ray(const float o[3], const float pg[3], int is_clearing, int carving, float max_ray_length_m,
                    float voxel_size_inv, float truncation, int cast_from_origin, int64_t* out, size_t cap) {
  RayCaster rc(Point(o[0], o[1], o[2]), Point(pg[0], pg[1], pg[2]), is_clearing != 0, carving != 0,
               max_ray_length_m, voxel_size_inv, truncation, cast_from_origin != 0);
  size_t n = 0;
  GlobalIndex g;
  while (rc.nextRayIndex(&g)) {
    if (n < cap) { out[3 * n] = g.x(); out[3 * n + 1] = g.y(); out[3 * n + 2] = g.z(); }
    ++n;
  }
  return n;
}

orc_approx_set* orc_approx_set_create(int small) {
  auto* r = new orc_approx_set;
  if (small) r->s.reset(new ApproxSetImpl<16, 10>()); else r->s.reset(new ApproxSetImpl<20, 10000>());
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
    off[3 * i] = NeighborhoodLookupTables::kOffsets(0, i);
    off[3 * i + 1] = NeighborhoodLookupTables::kOffsets(1, i);
    off[3 * i + 2] = NeighborhoodLookupTables::kOffsets(2, i);
    dist[i] = NeighborhoodLookupTables::kDistances(0, i);
  }
}

}  // extern "C"
