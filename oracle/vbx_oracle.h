/* ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * C API over the CPU restatement of the voxblox TSDF/ESDF hot path
 * (oracle/vbx_core.hpp, vbx_tsdf.hpp, vbx_esdf.hpp).  Loaded with ctypes by
 * tests/, by bench.py's cpu_baseline leg and by __graft_entry__.smoke() as the
 * CHECKER.  The product library (voxblox_amd/csrc, include/vbx_hip.h) never
 * links or calls anything declared here.
 */
#ifndef VBX_ORACLE_H_
#define VBX_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* POD mirror of TsdfIntegratorBase::Config (tsdf_integrator.h:56-89) plus the
 * two oracle-only switches documented in vbx_tsdf.hpp. */
typedef struct orc_tsdf_cfg {
  float default_truncation_distance;
  float max_weight;
  int32_t voxel_carving_enabled;
  float min_ray_length_m;
  float max_ray_length_m;
  int32_t use_const_weight;
  int32_t allow_clear;
  int32_t use_weight_dropoff;
  int32_t use_sparsity_compensation_factor;
  float sparsity_compensation_factor;
  int32_t integrator_threads;
  int32_t integration_order_mode; /* 0 "mixed", 1 "sorted" */
  int32_t enable_anti_grazing;
  float start_voxel_subsampling_factor;
  int32_t max_consecutive_ray_collisions;
  int32_t clear_checks_every_n_frames;
  float max_integration_time_s;
  int32_t oracle_merged_sorted_bundles;
  int32_t oracle_fast_exact_observed_set;
} orc_tsdf_cfg;

/* POD mirror of EsdfIntegrator::Config (esdf_integrator.h:29-78). */
typedef struct orc_esdf_cfg {
  int32_t full_euclidean_distance;
  float max_distance_m;
  float min_distance_m;
  float default_distance_m;
  float min_diff_m;
  float min_weight;
  int32_t num_buckets;
  int32_t multi_queue;
  int32_t add_occupied_crust;
  float clear_sphere_radius;
  float occupied_sphere_radius;
  int32_t oracle_orderfree_sign_mismatch; /* oracle-only switch, see vbx_esdf.hpp */
  int32_t oracle_unrestricted_wavefront;  /* oracle-only switch, see vbx_esdf.hpp */
} orc_esdf_cfg;

void orc_tsdf_cfg_default(orc_tsdf_cfg* cfg);
void orc_esdf_cfg_default(orc_esdf_cfg* cfg);

typedef struct orc_map orc_map; /* a TSDF layer + an ESDF layer of equal geometry */

orc_map* orc_map_create(float voxel_size, uint32_t voxels_per_side);
void orc_map_destroy(orc_map* m);

/* kind: 1 simple, 2 merged, 3 fast (TsdfIntegratorType, tsdf_integrator.h:30-34). */
typedef struct orc_tsdf_integrator orc_tsdf_integrator;
orc_tsdf_integrator* orc_tsdf_integrator_create(orc_map* m, int kind, const orc_tsdf_cfg* cfg);
void orc_tsdf_integrator_destroy(orc_tsdf_integrator* it);
/* points_C: n x 3 floats (12 B stride); rgba: n x 4 bytes. */
int orc_tsdf_integrate(orc_tsdf_integrator* it, const float pos[3], const float quat_wxyz[4],
                       const float* points_C, const uint8_t* rgba, size_t n, int freespace);
/* out[0] voxel updates, out[1] rays cast, out[2]/[3] merged bundles / clear bundles of the last call */
void orc_tsdf_stats(orc_tsdf_integrator* it, uint64_t out[4], int reset);
/* Resets the process-global reset counter of the Fast integrator (tsdf_integrator.cc:564). */
void orc_fast_reset_counter_set(int64_t v);

typedef struct orc_esdf_integrator orc_esdf_integrator;
orc_esdf_integrator* orc_esdf_integrator_create(orc_map* m, const orc_esdf_cfg* cfg);
void orc_esdf_integrator_destroy(orc_esdf_integrator* it);
void orc_esdf_update_from_tsdf_layer(orc_esdf_integrator* it, int clear_updated_flag);
void orc_esdf_update_from_tsdf_layer_batch(orc_esdf_integrator* it);
/* EsdfIntegrator::updateFromTsdfBlocks(list, incremental) (:124-302) and clear() (esdf_integrator.h:138-142) */
void orc_esdf_update_from_tsdf_blocks(orc_esdf_integrator* it, const int32_t* idx_xyz, size_t n, int incremental);
void orc_esdf_integrator_clear(orc_esdf_integrator* it);
/* EsdfIntegrator::addNewRobotPosition, esdf_integrator.cc:25-92 */
void orc_esdf_add_new_robot_position(orc_esdf_integrator* it, const float position[3]);
/* out: lower, raise, new, raised, open_pops, relaxations, blocks */
void orc_esdf_stats(orc_esdf_integrator* it, uint64_t out[7], int reset);

/* layer: 0 tsdf, 1 esdf.  Block indices come out in unordered_map iteration order. */
size_t orc_num_blocks(orc_map* m, int layer);
size_t orc_block_indices(orc_map* m, int layer, int32_t* idx_xyz, size_t cap);
/* TSDF block -> SoA copies (vps^3 each); any output pointer may be NULL.  Returns 0 if absent. */
int orc_tsdf_block_get(orc_map* m, const int32_t idx[3], float* dist, float* weight,
                       uint8_t* rgba, uint8_t* updated_bits);
int orc_esdf_block_get(orc_map* m, const int32_t idx[3], float* dist, uint8_t* flags,
                       int32_t* parent_xyz, uint8_t* updated_bits);
/* Overwrite / create a TSDF block from SoA arrays (for loadMap-style tests). */
int orc_tsdf_block_set(orc_map* m, const int32_t idx[3], const float* dist, const float* weight,
                       const uint8_t* rgba, uint8_t updated_bits);
void orc_remove_distant_blocks(orc_map* m, int layer, const float center[3], double max_distance);
void orc_clear(orc_map* m, int layer);
/* libvbxref_hip.so only (the HIP drop-in behind voxblox's headers): blocks the drop-in has uploaded to / removed
 * from the device map of this map's TSDF layer while reconciling host-side Layer edits; zeros elsewhere. */
void orc_dropin_stats(orc_map* m, uint64_t out[2]);
/* counts voxels with weight > 1e-6 (evaluation_utils.cc:75-78 "observed") */
uint64_t orc_tsdf_count_observed(orc_map* m);

/* Block::serializeToIntegers / deserializeFromIntegers (src/core/block.cc).  serialize returns
 * the number of words (3 or 2 per voxel) or 0 if the block is absent; deserialize allocates the
 * block if needed, sets all updated bits like Layer::addBlockFromProto (layer_inl.h:227), and
 * returns 0 on a size mismatch. */
size_t orc_block_serialize(orc_map* m, int layer, const int32_t idx[3], uint32_t* words, size_t cap);
int orc_block_deserialize(orc_map* m, int layer, const int32_t idx[3], const uint32_t* words, size_t n);
/* write an ESDF block from SoA arrays (tests of the serialisation corner cases) */
int orc_esdf_block_set(orc_map* m, const int32_t idx[3], const float* dist, const uint8_t* flags,
                       const int32_t* parent_xyz, uint8_t updated_bits);

/* ---- MeshIntegrator<TsdfVoxel> over the map's TSDF layer (mesh_integrator.h) ---- */
typedef struct orc_mesh_layer orc_mesh_layer; /* MeshLayer(block_size) + the layer it meshes */
orc_mesh_layer* orc_mesh_layer_create(orc_map* m);
void orc_mesh_layer_destroy(orc_mesh_layer* ml);
/* MeshIntegrator(cfg{use_color, min_weight, integrator_threads}, &tsdf, &mesh).generateMesh(...) */
void orc_mesh_generate(orc_mesh_layer* ml, int use_color, float min_weight, int threads,
                       int only_mesh_updated_blocks, int clear_updated_flag);
size_t orc_mesh_num_blocks(orc_mesh_layer* ml);
size_t orc_mesh_block_indices(orc_mesh_layer* ml, int32_t* idx_xyz, size_t cap);
/* out = {#vertices, #normals, #colors, #indices, updated}; returns 0 if the mesh is absent */
int orc_mesh_block_sizes(orc_mesh_layer* ml, const int32_t idx[3], uint64_t out[5]);
int orc_mesh_block_get(orc_mesh_layer* ml, const int32_t idx[3], float* vertices, float* normals,
                       uint8_t* rgba, uint64_t* indices);
void orc_mesh_clear_updated(orc_mesh_layer* ml); /* mesh->updated = false everywhere (the publisher's job) */

/* ---- known-answer helpers (restate test_tsdf_map / test_approx_hash_array / test_bucket_queue) ---- */
void orc_grid_index_from_point(const float p[3], float grid_size_inv, int64_t out[3]);
void orc_center_point_from_grid_index(const int64_t idx[3], float grid_size, float out[3]);
void orc_origin_point_from_grid_index(const int32_t idx[3], float grid_size, float out[3]);
void orc_grid_index_from_origin_point(const float p[3], float grid_size_inv, int32_t out[3]);
void orc_block_index_from_global(const int64_t g[3], float vps_inv, int32_t out[3]);
void orc_local_from_global(const int64_t g[3], int vps, int32_t out[3]);
void orc_global_from_block_and_local(const int32_t b[3], const int32_t v[3], int vps, int64_t out[3]);
uint64_t orc_linear_index(const int32_t v[3], int vps);
void orc_voxel_index_from_linear(uint64_t lin, int vps, int32_t out[3]);
uint64_t orc_any_index_hash(const int32_t idx[3]);
uint64_t orc_long_index_hash(const int64_t idx[3]);
uint64_t orc_mixed_index(uint64_t seq, uint64_t n);
uint32_t orc_blend_two_colors(uint32_t rgba1, float w1, uint32_t rgba2, float w2);
void orc_transform_point(const float pos[3], const float quat_wxyz[4], const float p[3], float out[3]);
/* Casts the ray exactly as RayCaster does; returns the number of indices written (<= cap). */
size_t orc_cast_ray(const float origin[3], const float point_G[3], int is_clearing,
                    int voxel_carving, float max_ray_length_m, float voxel_size_inv,
                    float truncation, int cast_from_origin, int64_t* out_xyz, size_t cap);

typedef struct orc_approx_set orc_approx_set; /* small != 0: ApproxHashSet<16,10> (the reference test's size), else <20,10000> */
orc_approx_set* orc_approx_set_create(int small);
void orc_approx_set_destroy(orc_approx_set* s);
int orc_approx_set_replace_hash(orc_approx_set* s, uint64_t hash);
int orc_approx_set_is_present(orc_approx_set* s, uint64_t hash);
void orc_approx_set_reset(orc_approx_set* s);

typedef struct orc_bucket_queue orc_bucket_queue; /* BucketQueue<size_t> */
orc_bucket_queue* orc_bucket_queue_create(int num_buckets, double max_val);
void orc_bucket_queue_destroy(orc_bucket_queue* q);
void orc_bucket_queue_push(orc_bucket_queue* q, uint64_t key, double value);
uint64_t orc_bucket_queue_front(orc_bucket_queue* q);
void orc_bucket_queue_pop(orc_bucket_queue* q);
int orc_bucket_queue_empty(orc_bucket_queue* q);

void orc_neighbor_lut(int32_t offsets_xyz[78], float distances[26]);

#ifdef __cplusplus
}
#endif
#endif /* VBX_ORACLE_H_ */
