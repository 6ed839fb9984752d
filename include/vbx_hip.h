/* vbx_hip.h — C-ABI of libvbx_hip.so, the MI355X (gfx950) TSDF/ESDF integration hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, int status codes, no C++ or torch
 * types.  voxblox has no FFI of its own for this path — the "operator API" is the C++ class
 * surface of /root/reference/voxblox/include/voxblox/integrator/{tsdf,esdf}_integrator.h — so
 * each entry point below names the reference member it stands in for.  A C++ shim with the
 * reference's class names and signatures on top of this ABI is in voxblox_amd/host/ and the
 * binding a voxblox maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every call returns VBX_OK (0) or a negative error; vbx_last_error() gives the text.  The
 *     C++ shim turns a non-zero status into the reference's abort-on-error convention
 *     (glog CHECK / LOG(FATAL), tsdf_integrator.cc:11,22,29,41,247).
 *   - one handle = one (Layer<TsdfVoxel>, Layer<EsdfVoxel>) pair resident in HBM; calls on a
 *     handle come from one host thread at a time (integratePointCloud is documented
 *     "NOT thread safe", tsdf_integrator.h:94-95).
 *   - there is NO CPU fallback: without a HIP device vbx_create() fails.
 */
#ifndef VBX_HIP_H_
#define VBX_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VBX_OK 0
#define VBX_ERR_INVALID (-1)   /* bad argument */
#define VBX_ERR_HIP (-2)       /* HIP runtime error */
#define VBX_ERR_CAPACITY (-3)  /* block pool / hash map full */
#define VBX_ERR_UNSUPPORTED (-4)

/* Layer geometry — Layer<V>::Layer(voxel_size, voxels_per_side), core/layer.h:34-44. */
typedef struct vbx_map_cfg {
  float voxel_size;
  uint32_t voxels_per_side; /* power of two, 4..32 (reference default 16) */
  uint32_t max_blocks;      /* initial capacity of the HBM block pool; 0 = default (65536).  The pool doubles
                               when a call needs more blocks (Layer::allocateBlockPtrByIndex never fails,
                               layer.h:133-160) up to vbx_set_pool_limit / 2^32 voxel ids / device memory */
} vbx_map_cfg;

/* TsdfIntegratorBase::Config, tsdf_integrator.h:56-89 (same fields, same defaults via
 * vbx_tsdf_cfg_default).  integrator_threads is accepted and ignored (the GPU result equals
 * the reference's 1-thread result); integration_order_mode: 0 "mixed", 1 "sorted". */
typedef struct vbx_tsdf_cfg {
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
  int32_t integration_order_mode;
  int32_t enable_anti_grazing;
  float start_voxel_subsampling_factor;
  int32_t max_consecutive_ray_collisions;
  int32_t clear_checks_every_n_frames;
  float max_integration_time_s; /* Fast only (tsdf_integrator.cc:496-499): <= 0 integrates nothing, like the
                                   reference.  A positive budget: the device cannot look at the clock inside a
                                   frame, so the call takes the prefix of the reference's taking order
                                   (ThreadSafeIndex) that the budget pays for at the time per point measured on
                                   the handle's earlier calls (the first call takes everything);
                                   vbx_counters.points_taken / time_budget_exceeded say what was done */
  /* Not in the reference Config.  MergedTsdfIntegrator visits its ray bundles in the iteration
   * order of a std::unordered_map (tsdf_integrator.cc:440-456), which the clamped fold makes
   * observable.  0 (default): that order, reconstructed from libstdc++'s bucket-count schedule and
   * node placement rules (bit-exact; ~0.2 ms of host work per frame at 640x480, checked against
   * the container by vbx_selftest_unordered_order);
   * 1: ascending voxel key (no host step; same voxels, distances differ in the order-sensitive
   * ~1 % of them). */
  int32_t merged_bundle_order;
  /* Not in the reference Config.  FastTsdfIntegrator stops a ray after more than
   * max_consecutive_ray_collisions voxels that its voxel_observed_approx_set_ reports as already
   * seen; that set is a lossy 2^20-slot ApproxHashSet (voxels sharing a slot evict each other).
   * 0 (default): the reference's set, replayed exactly (bit-exact; an iterative replay on top of
   * the exact-set solve); 1: an exact voxel set (one solve, ~2x faster; a ray then stops where
   * the reference's would if its set had no evictions — ~1 % of the voxels differ). */
  int32_t fast_observed_set;
} vbx_tsdf_cfg;

/* EsdfIntegrator::Config, esdf_integrator.h:29-78. */
typedef struct vbx_esdf_cfg {
  int32_t full_euclidean_distance; /* needs max_distance_m / voxel_size < 120 (int8 parent vectors) */
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
  /* Not in the reference Config.  1 (default since round 5, like every other place where the reference's 1-thread result
   * depends on an implementation detail): the reference's OWN result — updateFromTsdfBlocks' voxel walk, the FIFO raise
   * queue, BucketQueue pop order with num_buckets / multi_queue, min_diff_m gating, updateVoxelFromNeighbors incl. its
   * unscaled LUT distance, the sign-mismatch rule as written (esdf_integrator.cc:124-530, bucket_queue.h:41-80) — replayed
   * in parallel, thousands of pops at a time, with the reference's bits as the result (~24 ms per update on
   * the 640x480 / 0.05 m stream where the reference build needs ~78 ms on one core of the same box).  The blocks are
   * visited in the order of the list given to vbx_esdf_update_blocks; vbx_esdf_update visits them in the iteration order
   * the reference's Layer would have (Layer::getAllUpdatedBlocks over its unordered_map, layer.h:194-203): the library
   * replays that container from the sequence in which the integrators hand blocks to the Layer
   * (vbx_block_indices_layer_order) — so the plain call reproduces a single-threaded reference run bit for bit AS LONG AS
   * the library saw every block join the Layer (integrate calls, vbx_blocks_upload).  Blocks of unknown provenance
   * (vbx_blocks_merge_sums on a sharded persistent map, vbx_blocks_deserialize, blocks integrated while
   * vbx_set_block_order_tracking was off, a log that overflowed) are walked in ascending (z,y,x) order behind the others:
   * the call then still succeeds, vbx_counters.esdf_order_inexact says how many blocks that concerned and stderr carries
   * one line per handle; pass the order yourself (vbx_esdf_update_blocks) when it matters.
   * COST OF THIS DEFAULT: latency ~100x the fast mode's (tens of ms per update instead of 0.3); device memory ~1.3 GB
   * of replay pools per handle whatever the size of the map (1.07 GB of it the targets' event lists) + ~21 B per pool
   * voxel for the queue arena and the target / hazard maps (64 B with multi_queue) = ~2.3 GB for a 20 k-block map,
   * allocated at the first reference-order update (or by vbx_esdf_reserve) and kept until vbx_destroy.  Handles that never run a
   * reference-order update allocate none of it.
   * 0: the fast mode — order-free wavefronts run to their exact fixed points on the whole chip (0.3 ms per update; NOT
   * the reference's result where it depends on the queue order: bit-exact for batch updates with min_diff_m = 0, inside
   * the reference's own min_diff_m envelope otherwise).
   * vbx_esdf_add_new_robot_position reads the field too: with 1 its raise_ / open_ pushes and updated_blocks_
   * insertions are kept in the reference's order for the next update, which must then run with 1 as well (and the
   * other way round: VBX_ERR_UNSUPPORTED when the two calls disagree). */
  int32_t reference_order;
} vbx_esdf_cfg;

/* MeshIntegratorConfig, mesh/mesh_integrator.h:47-66 (integrator_threads has no meaning here). */
typedef struct vbx_mesh_cfg {
  int32_t use_color;
  float min_weight;
} vbx_mesh_cfg;

void vbx_mesh_cfg_default(vbx_mesh_cfg* cfg);
void vbx_tsdf_cfg_default(vbx_tsdf_cfg* cfg);
void vbx_esdf_cfg_default(vbx_esdf_cfg* cfg);

typedef struct vbx_ctx vbx_ctx;

/* Layer<TsdfVoxel>/Layer<EsdfVoxel> construction (core/layer.h:34-44) on HIP device `device`. */
vbx_ctx* vbx_create(const vbx_map_cfg* cfg, int device);
void vbx_destroy(vbx_ctx* ctx);
const char* vbx_last_error(vbx_ctx* ctx); /* ctx may be NULL: error of the last failed vbx_create */

/* Layer::voxel_size() / voxels_per_side() (layer.h:205-211) and the pool's current capacity. */
int vbx_get_map_cfg(vbx_ctx* ctx, vbx_map_cfg* out);
/* Run all work of this handle on an existing HIP stream (e.g. torch's current stream);
 * NULL restores the handle's own stream. */
int vbx_set_stream(vbx_ctx* ctx, void* hip_stream);
/* Upper bound of the block pool's growth in blocks (0 = none but the 32-bit voxel ids and device memory).  A
 * call that needs more fails with VBX_ERR_CAPACITY, the map unchanged by it. */
int vbx_set_pool_limit(vbx_ctx* ctx, uint32_t max_blocks_limit);

/* TsdfIntegratorType, tsdf_integrator.h:30-34 */
#define VBX_TSDF_SIMPLE 1
#define VBX_TSDF_MERGED 2
#define VBX_TSDF_FAST 3

/* {Simple,Merged,Fast}TsdfIntegrator::integratePointCloud(T_G_C, points_C, colors,
 * freespace_points) — tsdf_integrator.h:100-103, tsdf_integrator.cc:242-267, 307-338, 555-590.
 * T_G_C is passed as translation + unit quaternion (w,x,y,z) (kindr QuatTransformation).
 * points_C: n x 3 float (12-byte stride, i.e. Pointcloud::data()); rgba: n x 4 bytes
 * (Colors::data()).  Host pointers; returns after the map in HBM has been updated. */
int vbx_tsdf_integrate(vbx_ctx* ctx, int kind, const vbx_tsdf_cfg* cfg, const float T_G_C_pos[3],
                       const float T_G_C_quat_wxyz[4], const float* points_C, const uint8_t* rgba,
                       size_t n, int freespace_points);
/* Same, with points_C / rgba already resident in HBM (device pointers); asynchronous on the
 * handle's stream apart from the size read-backs the algorithm needs. */
int vbx_tsdf_integrate_device(vbx_ctx* ctx, int kind, const vbx_tsdf_cfg* cfg,
                              const float T_G_C_pos[3], const float T_G_C_quat_wxyz[4],
                              const float* d_points_C, const uint8_t* d_rgba, size_t n,
                              int freespace_points);

/* EsdfIntegrator::updateFromTsdfLayer(clear_updated_flag) (esdf_integrator.cc:104-122) when
 * batch == 0, EsdfIntegrator::updateFromTsdfLayerBatch() (:94-102) when batch != 0. */
int vbx_esdf_update(vbx_ctx* ctx, const vbx_esdf_cfg* cfg, int batch, int clear_updated_flag);
/* EsdfIntegrator::EsdfIntegrator (esdf_integrator.cc:7-21) is where the reference sets its queues up; the device side of
 * that is workspace: the ESDF layer's device arrays (9 B per voxel of the map's capacity) and, with cfg->reference_order,
 * the replay's pools (~1.5 GB, sized by the super-step, not by the map).  Optional — the first vbx_esdf_update /
 * vbx_esdf_update_blocks / vbx_esdf_add_new_robot_position allocates whatever is missing — but a first update that has to
 * allocate spends 30-100 ms in hipMalloc with the device idle.  Call it where the application constructs its integrator. */
int vbx_esdf_reserve(vbx_ctx* ctx, const vbx_esdf_cfg* cfg);

/* EsdfIntegrator::updateFromTsdfBlocks(tsdf_blocks, incremental) (esdf_integrator.cc:124-302): the
 * same update restricted to the listed TSDF blocks (blocks not in the TSDF layer are skipped); no
 * Update bit is cleared and the blocks addNewRobotPosition queued stay queued for classification. */
int vbx_esdf_update_blocks(vbx_ctx* ctx, const vbx_esdf_cfg* cfg, const int32_t* idx_xyz, size_t n,
                           int incremental);
/* EsdfIntegrator::clear() (esdf_integrator.h:138-142): forget the work addNewRobotPosition queued. */
int vbx_esdf_integrator_clear(vbx_ctx* ctx);

/* MeshIntegrator<TsdfVoxel>::generateMesh(only_mesh_updated_blocks, clear_updated_flag)
 * (mesh/mesh_integrator.h:142-195; per block updateMeshForBlock :250-270 = extractBlockMesh
 * :197-248 + MarchingCubes::meshCube marching_cubes.h:70-111 + updateMeshColor :372-392).  Meshes
 * the TSDF blocks carrying Update::kMesh (or every block) on the device and keeps the result of
 * THIS call device-resident until the next one: per meshed block a run of vertices (three per
 * triangle, in the reference's emission order), per-vertex normals and, with cfg->use_color,
 * per-vertex colours.  Mesh::indices is 0..n-1 per block (marching_cubes.h:94-96) and is not
 * stored.  A block without triangles is still listed (the reference clears its Mesh and sets
 * Mesh::updated).  The persistent MeshLayer (mesh_layer.h) stays with the caller. */
int vbx_mesh_generate(vbx_ctx* ctx, const vbx_mesh_cfg* cfg, int only_mesh_updated_blocks,
                      int clear_updated_flag, size_t* n_blocks, size_t* n_vertices);
/* The blocks meshed by the last vbx_mesh_generate: idx_xyz[3*i..] and vertex_offset[i]..[i+1]
 * (cap + 1 entries) into the vertex arrays.  *n is always set; VBX_ERR_CAPACITY if cap is short. */
int vbx_mesh_blocks(vbx_ctx* ctx, int32_t* idx_xyz, uint64_t* vertex_offset, size_t cap, size_t* n);
/* Copies the vertex arrays of the last vbx_mesh_generate to the host (any pointer may be NULL;
 * rgba = Color r,g,b,a per vertex). */
int vbx_mesh_download(vbx_ctx* ctx, float* vertices, float* normals, uint8_t* rgba, size_t cap_vertices);
/* The same arrays where they lie in device memory (valid until the next vbx_mesh_generate). */
int vbx_mesh_device_ptrs(vbx_ctx* ctx, const float** d_vertices, const float** d_normals,
                         const uint8_t** d_rgba);

/* EsdfIntegrator::addNewRobotPosition(position) (esdf_integrator.cc:25-92; caller
 * esdf_server.cc:219-226): unknown or hallucinated voxels within cfg->clear_sphere_radius become
 * free, remaining unknown voxels within cfg->occupied_sphere_radius occupied (both
 * "hallucinated"); ESDF blocks are allocated as needed.  The wavefront work it queues is done by
 * the next vbx_esdf_update(batch == 0); a batch update discards it with the layer.
 * cfg->reference_order = 1: the entries the reference pushes into raise_ (:48) and open_ (:84) are kept, in its
 * order — the iteration order of getSphereAroundPoint's HierarchicalIndexMap (planning_utils_inl.h:13-50,
 * common.h:97-99: an unordered_map with AnyIndexHash, block_hash.h:20-32, reproduced on the host with the same
 * insertions and the libstdc++ this library is built with) — and the next reference-order update starts from queues
 * that hold them; the blocks of updated_blocks_ (:54, :80) follow that update's TSDF blocks like in :107-109. */
int vbx_esdf_add_new_robot_position(vbx_ctx* ctx, const vbx_esdf_cfg* cfg, const float position[3]);
/* EsdfIntegrator::updated_blocks_ as the reference-order calls of vbx_esdf_add_new_robot_position left it:
 * order 0 = the sequence of first insertions (a caller that keeps its own IndexSet inserts them one by one),
 * order 1 = the iteration order of the reference's IndexSet.  Up to `cap` block indices (x, y, z) are written to
 * out_xyz (may be NULL), *n_out gets the size of the set; clear != 0 empties it (a caller that composes the list of
 * vbx_esdf_update_blocks itself, like updateFromTsdfLayer does :105-109; vbx_esdf_update appends and empties it on
 * its own). */
int vbx_esdf_robot_updated_blocks(vbx_ctx* ctx, int order, int32_t* out_xyz, size_t cap, size_t* n_out, int clear);

/* ---- host <-> HBM coherence for the callers that read/modify the Layer directly
 *      (mesher, publishers, removeDistantBlocks, load_map) ---- */
#define VBX_LAYER_TSDF 0
#define VBX_LAYER_ESDF 1
/* Update::Status bits, core/block.h:15-18 */
#define VBX_UPDATE_MAP 1
#define VBX_UPDATE_MESH 2
#define VBX_UPDATE_ESDF 4
/* Not an Update::Status bit.  ESDF layer only: "voxels changed on the device since vbx_clear_updated(…, VBX_UPDATE_DIRTY)".
 * The reference's wavefront writes neighbour blocks in place without flagging them (esdf_integrator.cc:305-496 only
 * flags the blocks it classifies, :147); a host mirror must take those too.  Accepted by vbx_blocks_updated and
 * vbx_clear_updated in update_mask; never reported in updated_bits. */
#define VBX_UPDATE_DIRTY 8

/* Layer::getNumberOfAllocatedBlocks / getAllAllocatedBlocks (layer.h:184-193, 205).  Indices
 * are returned in ascending (z,y,x) order. */
int vbx_num_blocks(vbx_ctx* ctx, int layer, size_t* n);
int vbx_block_indices(vbx_ctx* ctx, int layer, int32_t* idx_xyz, size_t cap, size_t* n);
/* Layer::getAllUpdatedBlocks(bit) (layer.h:194-203); update_mask = OR of VBX_UPDATE_*. */
int vbx_blocks_updated(vbx_ctx* ctx, int layer, int update_mask, int32_t* idx_xyz, size_t cap,
                       size_t* n);
/* The TSDF blocks the LAST vbx_tsdf_integrate[_device] call added to the Layer, in the sequence in which the reference's
 * single-threaded integrator inserts them: allocateStorageAndGetVoxelPtr emplaces a missing block in temp_block_map_ the
 * first time a ray reaches it (tsdf_integrator.cc:107-121) and updateLayerWithStoredBlocks walks that container into
 * Layer::insertBlock (:137-147; the Merged integrator does so once per pass, :329-336).  The device records every new
 * block's first touch in the reference's taking order; the host replays the container (same hash, same libstdc++, the
 * bucket array clear() leaves behind included).  A host Layer that allocates its new blocks in this sequence iterates
 * like the Layer of a CPU run (Layer::getAllUpdatedBlocks, layer.h:194-203) — which is what the ESDF's walk order, and
 * with it its result, depends on (esdf_integrator.cc:104-143). */
int vbx_blocks_new_ordered(vbx_ctx* ctx, int32_t* idx_xyz, size_t cap, size_t* n);
/* Layer::getAllAllocatedBlocks (update_mask 0) / getAllUpdatedBlocks(bit) of the TSDF layer in the iteration order the
 * reference's block_map_ would have, as far as the library saw the Layer being built (integrate calls in the sequence
 * above, vbx_blocks_upload in list order, removals); *exact (optional) = 0 when blocks of unknown provenance had to be
 * appended in ascending order.  vbx_esdf_update with cfg->reference_order walks this order. */
int vbx_block_indices_layer_order(vbx_ctx* ctx, int update_mask, int32_t* idx_xyz, size_t cap, size_t* n, int* exact);
/* on = 0: the map stops following the reference's insertion order (no first-touch ranks, no log; the integrators' emit
 * kernels then issue no atomics at all — the fold publishes a touched block from one thread).  For scratch maps whose block
 * order nobody asks for: the per-step delta maps of the sharding, where every block is new in every step
 * (libvbx_shard.so and voxblox_amd.multi_gpu switch it off for their deltas).  Default on.  Call it on a map without
 * blocks if vbx_block_indices_layer_order is to stay exact. */
int vbx_set_block_order_tracking(vbx_ctx* ctx, int on);
/* Block<V> voxel array in the reference's AoS layout: TsdfVoxel = {float distance; float
 * weight; uint8 r,g,b,a} (12 B, voxel.h:12-16); EsdfVoxel = {float distance; uint8 observed,
 * hallucinated, in_queue, fixed; int32 parent[3]} (20 B, voxel.h:18-37).  Returns
 * VBX_ERR_INVALID if the block is not allocated. */
int vbx_block_download(vbx_ctx* ctx, int layer, const int32_t idx[3], void* aos_voxels,
                       uint8_t* updated_bits, uint8_t* has_data);
/* The same for n blocks in one pack kernel + one device-to-host copy: aos_voxels receives n
 * consecutive voxel arrays, updated_bits / has_data (optional) n bytes each.  This is the call the
 * per-frame host mirror uses (the blocks Layer::getAllUpdatedBlocks(kMap) returns). */
int vbx_blocks_download(vbx_ctx* ctx, int layer, const int32_t* idx_xyz, size_t n, void* aos_voxels,
                        uint8_t* updated_bits, uint8_t* has_data);
/* Page-locked host memory for the mirror's staging buffer: device-to-host copies into it run at
 * link speed, copies into pageable memory are several times slower.  Any host pointer works. */
void* vbx_host_alloc(size_t bytes);
void vbx_host_free(void* p);
/* Layer::allocateBlockPtrByIndex + overwrite (load_map / tsdfMapCallback path, tsdf_server.cc:566-578,
 * 639-653).  vbx_blocks_upload takes n blocks in one call: idx_xyz n x 3, aos_voxels n x vps^3 voxels in the
 * reference's AoS layout (12 B TsdfVoxel / 20 B EsdfVoxel), one updated-bits byte and (TSDF) one has_data
 * byte per block; keys are inserted and slots assigned on the device, one staging copy, one unpack kernel. */
int vbx_block_upload(vbx_ctx* ctx, int layer, const int32_t idx[3], const void* aos_voxels,
                     uint8_t updated_bits, uint8_t has_data);
int vbx_blocks_upload(vbx_ctx* ctx, int layer, const int32_t* idx_xyz, size_t n, const void* aos_voxels,
                      const uint8_t* updated_bits, const uint8_t* has_data);
/* Layer::removeBlock / removeDistantBlocks / removeAllBlocks (layer.h:167-182).  A removed block frees its
 * memory like in the reference: once a pool slot holds no block of either layer its hash entry and the slot
 * are recycled, so max_blocks bounds the blocks ALIVE at one time, not the blocks ever touched (a sliding
 * window map keeps running); a map without any block left is back in its initial state. */
int vbx_block_remove(vbx_ctx* ctx, int layer, const int32_t idx[3]);
/* The same for n blocks with one pass over the pool (what a host loop mirroring Layer::removeBlock should call). */
int vbx_blocks_remove(vbx_ctx* ctx, int layer, const int32_t* idx_xyz, size_t n);
int vbx_remove_distant_blocks(vbx_ctx* ctx, int layer, const float center[3], double max_distance);
int vbx_clear(vbx_ctx* ctx, int layer);
/* vbx_clear(TSDF) for a scratch map that will see the same region again (the per-step delta maps of the multi-GPU
 * sharding): the blocks are zeroed and leave the layer — vbx_num_blocks is 0 afterwards, nothing is listed, exported or
 * downloaded — but their pool slots and hash entries stay as invisible candidates, so the next frame need not allocate
 * its blocks again.  Observable behaviour equals vbx_clear; the pool keeps the union of the blocks ever touched. */
int vbx_clear_keep_slots(vbx_ctx* ctx);
/* block.updated().reset(bit) over all blocks of a layer (mesher / ESDF consumers). */
int vbx_clear_updated(vbx_ctx* ctx, int layer, int update_mask);

/* ---- Layer serialization ----
 * Block<V>::serializeToIntegers / deserializeFromIntegers (src/core/block.cc:66-90, 112-137,
 * 160-183, 204-234): the uint32 word stream that both the .voxblox file format
 * (BlockProto.voxel_data, layer_inl.h:139-158) and voxblox_msgs/Block.data
 * (conversions_inl.h:34-38) carry.  TSDF: 3 words per voxel {bits(distance), bits(weight),
 * r<<24|g<<16|b<<8|a}; ESDF: 2 words {bits(distance), parent<<8|flags} including the reference
 * writer's sign-extension of negative parent components (block.cc:37-39).  Packed/unpacked on
 * the GPU straight from/to the SoA pool; n blocks per call, words_per_block = vps^3 * (3|2).
 * words: host buffer of n * words_per_block uint32.  Serializing an unallocated block is
 * VBX_ERR_INVALID.  Deserializing allocates, publishes and sets all Update bits
 * (Layer::addBlockFromProto, layer_inl.h:199-230); has_data[i] may be NULL (= 0). */
int vbx_blocks_serialize(vbx_ctx* ctx, int layer, const int32_t* idx_xyz, size_t n, uint32_t* words,
                         uint8_t* has_data);
int vbx_blocks_deserialize(vbx_ctx* ctx, int layer, const int32_t* idx_xyz, size_t n, const uint32_t* words,
                           const uint8_t* has_data);

/* ---- multi-GPU: ray-bundle sharding with a block merge ----
 * Each rank integrates its ray shard into a per-frame delta map; the deltas are combined as
 * weighted sums — which is what Block::mergeBlock / mergeVoxelAIntoVoxelB compute
 * (core/block_inl.h:112-129, src/utils/voxel_utils.cc:10-22): every touched block's sums go to
 * the rank that owns the block (a sparse RCCL all-to-all-v of the touched blocks only, grouped by owner; a
 * reduce-scatter over the union of touched blocks would move the same sums N times), and the owner adds
 * the senders' rows up and folds them into its shard of the persistent map. */
/* For each listed block writes one ROW of three planes of nvox = vps^3 32-bit words each — distance (float), weight
 * (float), colour (the four bytes r,g,b,a as one word) — layout d_out[(i*3 + plane)*nvox + linear_index], zeros for
 * blocks this map does not hold: the delta voxels themselves, 48 KiB per block at vps 16, what mergeVoxelAIntoVoxelB reads of
 * voxel A (voxel_utils.cc:10-22).  (Rounds 1-4 exported the six products [w*d, w, w*r, w*g, w*b, w*a], 96 KiB; the
 * products are now formed by vbx_blocks_merge_sums with the same float operations — the merged map is bit for bit the
 * same.)  idx_xyz is a host array, d_out a device pointer. */
int vbx_blocks_export_sums(vbx_ctx* ctx, const int32_t* idx_xyz, size_t n, float* d_out);
/* Folds rows (same layout, device pointer) into this map.  A BlockIndex may be listed more than once (several
 * senders touched the block): the weighted sums of its rows are added up first, in row order.  Then A = {d = Swd/Sw, w = Sw,
 * colour = round(Swc/Sw)} merged into the stored voxel B exactly as mergeVoxelAIntoVoxelB does
 * (d = (dA*wA + dB*wB)/(wA+wB), colour = blendTwoColors(A,wA,B,wB), w = wA+wB; nothing when
 * wA+wB <= 0).  Blocks are allocated as needed and get all Update bits.  If apply_caps != 0
 * the result is clamped like updateTsdfVoxel does (|d| <= truncation, w <= max_weight,
 * tsdf_integrator.cc:205-208), which mergeVoxelAIntoVoxelB itself does not do. */
int vbx_blocks_merge_sums(vbx_ctx* ctx, const int32_t* idx_xyz, size_t n, const float* d_sums,
                          int apply_caps, float truncation_distance, float max_weight);

/* ---- measurement ---- */
typedef struct vbx_counters {
  uint64_t points;          /* points handed to the last integrate call */
  uint64_t rays_cast;       /* rays actually cast */
  uint64_t voxel_updates;   /* updateTsdfVoxel evaluations */
  uint64_t voxels_touched;  /* distinct voxels updated (U in SURVEY §8(d)) */
  uint64_t blocks_allocated;/* blocks newly published by the last call */
  uint64_t iterations;      /* Fast: fixed-point sweeps of the early-termination solver */
  uint64_t esdf_blocks;     /* ESDF: TSDF blocks propagated */
  uint64_t esdf_relaxations;/* ESDF: successful wavefront relaxations */
  uint64_t esdf_sweeps;     /* ESDF: wavefront sweeps */
  uint64_t replay_rounds;   /* Fast, fast_observed_set = 0: rounds of the observed-set replay */
  uint64_t replay_block_rounds; /* ... of which: rounds run on blocks of consecutive rays (fine voxels) */
  uint64_t time_budget_exceeded; /* Fast: 1 if the last call outran cfg->max_integration_time_s or was cut short for it */
  uint64_t esdf_respeculated; /* ESDF: 1 if a phase needed more sweeps than were queued ahead of the read-back and the
                                 update was finished sweep by sweep (VBX_ESDF_RAISE_SWEEPS / VBX_ESDF_LOWER_SWEEPS) */
  uint64_t points_taken;    /* Fast: points of the taking order the last call took (= points unless
                               cfg->max_integration_time_s cut the frame short) */
  uint64_t esdf_order_inexact; /* ESDF, reference_order = 1, vbx_esdf_update: blocks the walk had to list in ascending
                               (z,y,x) order behind the ones whose place in the reference Layer's iteration order the
                               library knows (blocks merged in from another map, deserialised, integrated while
                               vbx_set_block_order_tracking was off, or behind an overflowed log).  0: the update walked
                               the blocks exactly like a single-threaded reference run */
} vbx_counters;
int vbx_get_counters(vbx_ctx* ctx, vbx_counters* out);

/* `static int64_t reset_counter` of FastTsdfIntegrator::integratePointCloud (tsdf_integrator.cc:564-569): the reference
 * counts the calls of ALL FastTsdfIntegrator instances of the process in one function-static and clears the two
 * ApproxHashSets of the instance whose call makes it reach clear_checks_every_n_frames.  The library keeps the same ONE
 * counter per process (per loaded libvbx_hip.so), shared by every handle on every device; with the default
 * clear_checks_every_n_frames = 1 it is invisible.  The reference offers no access to it; these two exist so that a test
 * (or a process that wants per-map behaviour) can put it into a known state. */
int64_t vbx_fast_reset_counter_get(void);
void vbx_fast_reset_counter_set(int64_t value);

/* Self-test hook (no reference counterpart): checks the library's stable radix sort — the primitive
 * under the start-voxel replay, the observed-set replay and the ordered fold — against
 * std::stable_sort on n pseudo-random keys, bit field [begin_bit, end_bit). */
int vbx_selftest_sort(vbx_ctx* ctx, uint32_t n, uint32_t begin_bit, uint32_t end_bit, uint32_t seed, int with_vals);
/* Self-test hook: the library's single-launch exclusive prefix sum (ray offsets, compaction, radix histograms,
 * probe offsets) against a host loop on n pseudo-random counters, `repeats` back-to-back calls. */
int vbx_selftest_scan(vbx_ctx* ctx, uint32_t n, uint32_t seed, uint32_t repeats);

/* Self-test hook, host only (no GPU needed): the Merged integrator's reconstruction of libstdc++'s
 * std::unordered_map iteration order (tsdf_integrator.cc:440-456, see vbx_host_tsdf.hpp) against the
 * container itself, on n distinct keys with pseudo-random hashes & hash_mask. */
int vbx_selftest_unordered_order(uint32_t n, uint32_t seed, uint32_t hash_mask);
/* Self-test hook, host only: the iteration order of the library's stand-in for the reference's IndexSet /
 * HierarchicalIndexMap (block_hash.h:20-48; reference-order addNewRobotPosition) after inserting the n block indices
 * of idx_xyz in sequence.  Writes the distinct indices in iteration order to out_xyz (room for n) and returns how
 * many there are (< 0 on bad arguments); tests compare it with the oracle's containers. */
int64_t vbx_selftest_index_set_order(const int32_t* idx_xyz, size_t n, int32_t* out_xyz);

/* HIP-event timing of the last integrate / esdf call on the handle's stream, in ms. */
typedef struct vbx_timing {
  float total_ms;
  float prep_ms;     /* validate/transform/bundle/dedupe */
  float alloc_ms;    /* block allocation walk */
  float solve_ms;    /* Fast: early-termination solver */
  float emit_ms;     /* ray march emitting ordered voxel updates */
  float sort_ms;     /* ordering of the updates */
  float fold_ms;     /* per-voxel ordered fold (the TSDF update itself) */
  float replay_ms;   /* Fast, fast_observed_set = 0: replay rounds of the reference's observed-voxel set */
} vbx_timing;
int vbx_enable_timing(vbx_ctx* ctx, int enable);
int vbx_get_timing(vbx_ctx* ctx, vbx_timing* out);

/* Per-kernel profile (measurement only, no reference counterpart; bench.py's roofline block is computed
 * from it).  While enabled, every kernel launch of the integrate / ESDF / mesh calls is bracketed by two
 * HIP events on the handle's stream; the table accumulates launches and milliseconds per kernel name
 * until vbx_profile_reset.  vbx_profile_get writes one line per kernel, "name\tlaunches\ttotal_ms\n"
 * (NUL-terminated, truncated to cap), the size a full copy needs to *needed and the number of API calls
 * profiled to *calls.  Costs ~2 us of host time per launch while on: not for the timed region. */
int vbx_profile_enable(vbx_ctx* ctx, int enable);
int vbx_profile_reset(vbx_ctx* ctx);
int vbx_profile_get(vbx_ctx* ctx, char* buf, size_t cap, size_t* needed, uint64_t* calls);

#ifdef __cplusplus
}
#endif
#endif /* VBX_HIP_H_ */
