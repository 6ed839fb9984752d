/* vbx_shard.h — C-ABI of the multi-GPU ray-bundle sharding layer (libvbx_shard.so).
 *
 * north_star / SURVEY 8(e): point clouds shard across the GPUs of one node by ray bundle, overlapping block
 * updates are combined over RCCL with the reference's own merge semantics — Block::mergeBlock /
 * mergeVoxelAIntoVoxelB (core/block_inl.h:112-129, src/utils/voxel_utils.cc:10-22) is a weighted sum.
 * One process per GPU.  Per time step every rank integrates ITS ray shards (whole sensors, or contiguous
 * bands of a cloud) into a zeroed per-step delta map; vbx_shard_end_step then sends every touched block of the
 * delta (distance, weight, colour: 48 KiB at vps 16) to the block's owner rank
 * (owner = hash(BlockIndex) mod world) with ONE sparse all-to-all-v — only touched blocks travel — and the owner
 * forms the partial sums (w*d, w, w*r, w*g, w*b, w*a), adds the rows of equal BlockIndex in (sender rank, key) order and folds the
 * result into its shard of the persistent map (vbx_blocks_merge_sums).  The persistent map is distributed
 * by block ownership; no rank holds all of it.
 *
 * This is the C++ host path a voxblox_ros node links (no Python, no torch): libvbx_shard.so = this file's
 * entry points over libvbx_hip.so and librccl.so.  voxblox_amd/multi_gpu.py is the same protocol over
 * torch.distributed (RCCL on GPUs, gloo in the CPU tests); tests/test_shard_native.py checks that both
 * produce the same map.  The reference has no multi-device path: there is no reference interface to cite
 * beyond the merge functions above.
 */
#ifndef VBX_SHARD_H_
#define VBX_SHARD_H_

#include <stddef.h>
#include <stdint.h>

#include "vbx_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define VBX_SHARD_ID_BYTES 128 /* = NCCL_UNIQUE_ID_BYTES */

typedef struct vbx_shard vbx_shard;

/* Rank 0 creates the communicator id (ncclGetUniqueId) and hands the bytes to the other ranks by whatever
 * means the application has (ROS parameter, file, MPI, torch.distributed broadcast ...). */
int vbx_shard_get_unique_id(uint8_t id[VBX_SHARD_ID_BYTES]);

/* persistent: this rank's shard of the map (blocks it owns); delta: the per-step scratch map (same geometry).
 * Both stay owned by the caller.  world == 1 needs no id (NULL: no RCCL call is made; with an id the
 * one-rank communicator is created and every collective runs for real — used by the tests on a 1-GPU box). */
vbx_shard* vbx_shard_create(vbx_ctx* persistent, vbx_ctx* delta, int rank, int world,
                            const uint8_t id[VBX_SHARD_ID_BYTES], int device);
void vbx_shard_destroy(vbx_shard* s);
const char* vbx_shard_last_error(vbx_shard* s);

/* One time step: begin (clears the delta map), any number of vbx_shard_integrate calls (same arguments as
 * vbx_tsdf_integrate_device, into the delta map), end (exchange + owner merge; collective: every rank of the
 * communicator must call it once per step, in the same order).  apply_caps as in vbx_blocks_merge_sums. */
int vbx_shard_begin_step(vbx_shard* s);
int vbx_shard_integrate(vbx_shard* s, int kind, const vbx_tsdf_cfg* cfg, const float pos[3], const float quat_wxyz[4],
                        const float* d_points_C, const uint8_t* d_rgba, size_t n, int freespace_points);
int vbx_shard_end_step(vbx_shard* s, int apply_caps, float truncation_distance, float max_weight);

/* More delta maps (same geometry; owned by the caller).  With n delta maps shard i of a step goes into delta i % n:
 * as many deltas as shards = every ray shard (a sensor, or a band of one) has a delta map of its own, the shards of a
 * step are integrated CONCURRENTLY (one host thread and one HIP stream per delta map — the integration is a chain of
 * small dependent launches that leaves most of the chip idle) and the merged map depends on the shard layout only,
 * not on how the shards were dealt to the ranks.  vbx_shard_integrate keeps integrating into delta 0. */
int vbx_shard_add_delta(vbx_shard* s, vbx_ctx* delta);
/* All shards of this rank's step at once: pos_xyz n x 3, quat_wxyz n x 4, device pointers and point counts per shard. */
int vbx_shard_integrate_shards(vbx_shard* s, int kind, const vbx_tsdf_cfg* cfg, size_t n_shards, const float* pos_xyz,
                               const float* quat_wxyz, const float* const* d_points_C, const uint8_t* const* d_rgba,
                               const size_t* n_points, int freespace_points);
/* on != 0: the registered delta maps (an even number) become two alternating sets; vbx_shard_end_step then starts the
 * exchange + owner merge on a worker thread and returns, so that it runs behind the next step's integration (the same
 * overlap voxblox_amd.multi_gpu.PipelinedShardedTsdfMap has); a failed exchange is reported by the next
 * vbx_shard_end_step or by vbx_shard_wait, which also must be called before the persistent map is read. */
int vbx_shard_set_pipelined(vbx_shard* s, int on);
int vbx_shard_wait(vbx_shard* s);

typedef struct vbx_shard_stats {
  uint64_t steps;
  uint64_t sent_blocks;      /* blocks this rank's deltas touched (= rows sent, own ones included) */
  uint64_t received_blocks;  /* rows this rank received as owner */
  uint64_t payload_bytes;    /* bytes of block rows sent */
} vbx_shard_stats;
int vbx_shard_get_stats(vbx_shard* s, vbx_shard_stats* out);

/* Owner rank of a BlockIndex (the same function as voxblox_amd.multi_gpu.owner_of). */
int vbx_shard_owner_of(const int32_t idx[3], int world);

#ifdef __cplusplus
}
#endif
#endif /* VBX_SHARD_H_ */
