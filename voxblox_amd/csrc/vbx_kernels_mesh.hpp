// Incremental mesher, SURVEY §8(f) #4: MeshIntegrator<TsdfVoxel>::generateMesh
// (mesh_integrator.h:142-392) + MarchingCubes::meshCube (marching_cubes.h:70-161).
//
// One workgroup per selected TSDF block.  The block's distances and the one-voxel shell it
// shares with its seven +x/+y/+z neighbours are staged in LDS as a (VPS+1)^3 tile (NaN = corner
// not usable: weight <= min_weight or block absent), every thread owns a few consecutive
// cubes *in the reference's emission order* (interior x-outer/z-inner, then the max-X, max-Y
// and max-Z planes, mesh_integrator.h:197-248), and a workgroup scan turns the per-cube triangle
// counts into output positions, so a block's vertex list is the reference's, element by element.
// Pass 1 (EMIT = false) only writes the block's triangle total; a device scan over the blocks
// gives each its slice of the output pool; pass 2 (EMIT = true) recomputes, queues the triangles
// as work items and writes one triangle per thread.
#pragma once
#include "vbx_mc_table.hpp"

namespace {

// threads per workgroup: an incremental update meshes ~100 blocks on 256 CUs, so the time is one
// workgroup's latency; 1024 threads (4 cubes each at VPS = 16) cut it 3x against 256
template <int VPS> struct MeshThreads { static constexpr int value = VPS >= 16 ? 1024 : 256; };

struct MeshDev {  // by-value kernel argument
  const uint32_t* list;   // selected pool slots
  const uint32_t* n_list; // device word: number of selected slots
  uint32_t* tri_count;    // [>= n_list + 1] pass 1 output
  const uint32_t* tri_off;  // exclusive scan of tri_count
  float* verts;           // 3 floats per vertex, 3 vertices per triangle
  float* normals;
  uint32_t* colors;       // rgba per vertex, or null (use_color = false)
  float min_weight;
  float block_size;
  float block_size_inv;
};

// slot 1 if the block is selected for meshing: part of the Layer, and (only_updated) carries
// Update::kMesh (mesh_integrator.h:149-153)
__global__ void k_mesh_select(MapDev m, uint32_t used, int only_updated, uint32_t* head) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > used) return;
  uint32_t sel = 0;
  if (s < used) {
    const uint32_t f = m.blk_flags[s];
    sel = (f & kFlagPublished) && (!only_updated || (f & 2u)) ? 1u : 0u;
  }
  head[s] = sel;
}
__global__ void k_mesh_compact(uint32_t used, const uint32_t* __restrict__ head, const uint32_t* __restrict__ rank,
                               uint32_t* list, uint32_t* tri_count) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > used) return;
  tri_count[s] = 0;
  if (s < used && head[s]) list[rank[s]] = s;
}
// block table for the host + the optional reset of Update::kMesh (mesh_integrator.h:189-193)
__global__ void k_mesh_finish(MapDev m, MeshDev d, int clear_flag, int32_t* out_idx, uint32_t* out_off) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n = *d.n_list;
  if (i > n) return;
  out_off[i] = d.tri_off[i];
  if (i == n) return;
  const uint32_t slot = d.list[i];
  out_idx[3 * i] = m.blk_idx[3 * slot];
  out_idx[3 * i + 1] = m.blk_idx[3 * slot + 1];
  out_idx[3 * i + 2] = m.blk_idx[3 * slot + 2];
  if (clear_flag) atomicAnd(&m.blk_flags[slot], ~2u);
}

// cube corner i sits at (x, y, z) = ((i ^ i >> 1) & 1, i >> 1 & 1, i >> 2) (mesh_integrator.h:98-99)
__device__ inline int mc_corner_x(int i) { return (i ^ (i >> 1)) & 1; }
__device__ inline int mc_corner_y(int i) { return (i >> 1) & 1; }
__device__ inline int mc_corner_z(int i) { return i >> 2; }
// cube edge e joins corners (marching_cubes.cc:288-290)
__device__ inline int mc_edge_c0(int e) { return e < 8 ? e : e - 8; }
__device__ inline int mc_edge_c1(int e) { return e < 4 ? ((e + 1) & 3) : e < 8 ? 4 + ((e + 1) & 3) : e - 4; }

// emission rank -> voxel, mesh_integrator.h:197-248
template <int VPS>
__device__ inline void mesh_rank_to_voxel(int r, int* x, int* y, int* z) {
  constexpr int V1 = VPS - 1, I = V1 * V1 * V1, PX = VPS * VPS, PY = VPS * V1;
  if (r < I) {
    *x = r / (V1 * V1);
    *y = (r / V1) % V1;
    *z = r % V1;
  } else if (r < I + PX) {
    r -= I;
    *x = V1; *z = r / VPS; *y = r % VPS;
  } else if (r < I + PX + PY) {
    r -= I + PX;
    *y = V1; *z = r / V1; *x = r % V1;
  } else {
    r -= I + PX + PY;
    *z = V1; *y = r / V1; *x = r % V1;
  }
}

template <int VPS, bool EMIT>
__global__ void __launch_bounds__(MeshThreads<VPS>::value) k_mesh_block(MapDev m, MeshDev d) {
  constexpr int kMeshThreads = MeshThreads<VPS>::value;
  constexpr int T = VPS + 1, NT = T * T * T, NV = VPS * VPS * VPS;
  constexpr int RPT = NV / kMeshThreads;  // cubes per thread, consecutive in emission order
  static_assert(NV % kMeshThreads == 0, "block size");
  __shared__ float sdf[NT];
  __shared__ uint32_t nslot[8];
  __shared__ uint32_t wsum[64];
  if (blockIdx.x >= *d.n_list) return;
  const uint32_t slot = d.list[blockIdx.x];
  const int tid = threadIdx.x;
  const int bx = m.blk_idx[3 * slot], by = m.blk_idx[3 * slot + 1], bz = m.blk_idx[3 * slot + 2];
  if (tid < 8) {
    uint32_t s = slot;
    if (tid) {
      s = map_find(m, pack_block_key(bx + (tid & 1), by + ((tid >> 1) & 1), bz + (tid >> 2)));
      if (s != kInvalidSlot && !(m.blk_flags[s] & kFlagPublished)) s = kInvalidSlot;  // Layer::hasBlock
    }
    nslot[tid] = s;
  }
  __syncthreads();
  for (int i = tid; i < NT; i += kMeshThreads) {
    const int x = i % T, y = (i / T) % T, z = i / (T * T);
    const uint32_t s = nslot[(x == VPS ? 1 : 0) | (y == VPS ? 2 : 0) | (z == VPS ? 4 : 0)];
    float v = __builtin_nanf("");
    if (s != kInvalidSlot) {
      const size_t a = (size_t)s * NV + (size_t)((x & (VPS - 1)) + VPS * ((y & (VPS - 1)) + VPS * (z & (VPS - 1))));
      if (!(m.weight[a] <= d.min_weight)) v = m.dist[a];  // utils::getSdfIfValid, meshing_utils.h:15-24
    }
    sdf[i] = v;
  }
  __syncthreads();

  // per cube: configuration and triangle count
  uint8_t cfg[RPT];
  uint32_t mine = 0;
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    int x, y, z;
    mesh_rank_to_voxel<VPS>(tid * RPT + k, &x, &y, &z);
    int c = 0;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float v = sdf[(x + mc_corner_x(i)) + T * ((y + mc_corner_y(i)) + T * (z + mc_corner_z(i)))];
      ok = ok && (v == v);
      c |= (v < 0.0f) ? (1 << i) : 0;  // calculateVertexConfiguration, marching_cubes.h:113-123
    }
    c = ok ? c : 0;
    cfg[k] = (uint8_t)c;
    mine += (uint32_t)(vbx_mc::kMcTriTable[c] >> 60);
  }
  // exclusive scan of the per-thread totals: shuffles inside each wave, the wave totals through LDS
  constexpr int NW = kMeshThreads / 64;
  const int lane = tid & 63, wave = tid >> 6;
  uint32_t incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t up = __shfl_up(incl, o, 64);
    if (lane >= o) incl += up;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  if (tid < 64) {
    uint32_t w = tid < NW ? wsum[tid] : 0;
#pragma unroll
    for (int o = 1; o < NW; o <<= 1) {
      const uint32_t up = __shfl_up(w, o, 64);
      if (tid >= o) w += up;
    }
    if (tid < NW) wsum[tid] = w;
  }
  __syncthreads();
  if (!EMIT) {
    if (tid == 0) d.tri_count[blockIdx.x] = wsum[NW - 1];
    return;
  }
  // Surface cubes are a few percent of a block and sit in a few threads: a thread emitting its own
  // cubes' triangles serially leaves the rest of the workgroup idle (46 us per pass).  Instead the
  // triangles become work items (cube rank << 3 | triangle number, at the block-local triangle
  // index the scan assigned), queued in LDS kItems at a time, and every thread emits one triangle
  // per step.
  constexpr int kItems = 4096;
  __shared__ uint16_t s_item[kItems];
  const uint32_t local0 = (wave ? wsum[wave - 1] : 0u) + incl - mine;  // first triangle of this thread's cubes
  const uint32_t block_tris = wsum[NW - 1];
  const uint32_t tri_base = d.tri_off[blockIdx.x];
  const f3 origin = {(float)bx * d.block_size, (float)by * d.block_size, (float)bz * d.block_size};  // layer.h:136-139
  for (uint32_t q0 = 0; q0 < block_tris; q0 += kItems) {
    {
      uint32_t at = local0;
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const int n_tri_k = (int)(vbx_mc::kMcTriTable[cfg[k]] >> 60);
        for (int t = 0; t < n_tri_k; ++t, ++at)
          if (at >= q0 && at < q0 + kItems) s_item[at - q0] = (uint16_t)(((tid * RPT + k) << 3) | t);
      }
    }
    __syncthreads();
    const uint32_t n_items = min((uint32_t)kItems, block_tris - q0);
    for (uint32_t it = tid; it < n_items; it += kMeshThreads) {
    const uint32_t item = s_item[it];
    const int rank_c = (int)(item >> 3), t = (int)(item & 7u);
    const uint32_t tri = tri_base + q0 + it;
    int x, y, z;
    mesh_rank_to_voxel<VPS>(rank_c, &x, &y, &z);
    int c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      c |= (sdf[(x + mc_corner_x(i)) + T * ((y + mc_corner_y(i)) + T * (z + mc_corner_z(i)))] < 0.0f) ? (1 << i) : 0;
    const uint64_t row = vbx_mc::kMcTriTable[c];
    // Block::computeCoordinatesFromVoxelIndex (block.h:90-92), then the eight corner positions
    // coords + offset * voxel_size (mesh_integrator.h:277-290)
    const f3 coords = f3_add(origin, center_point_from_grid_index(l3{x, y, z}, m.voxel_size));
    auto corner = [&](int i) -> f3 {
      return f3_add(coords, f3{(float)mc_corner_x(i) * m.voxel_size, (float)mc_corner_y(i) * m.voxel_size,
                               (float)mc_corner_z(i) * m.voxel_size});
    };
    auto corner_sdf = [&](int i) -> float {
      return sdf[(x + mc_corner_x(i)) + T * ((y + mc_corner_y(i)) + T * (z + mc_corner_z(i)))];
    };
    // MarchingCubes::interpolateVertex, marching_cubes.h:148-161
    auto edge_vertex = [&](int e) -> f3 {
      const int c0 = mc_edge_c0(e), c1 = mc_edge_c1(e);
      const f3 v1 = corner(c0), v2 = corner(c1);
      const float s1 = corner_sdf(c0), s2 = corner_sdf(c1);
      const float diff = s1 - s2;
      if (fabsf(diff) >= 1e-6f) {
        const float t = s1 / diff;
        return f3_add(v1, f3_mul(f3_sub(v2, v1), t));
      }
      return f3_mul(f3_add(v1, v2), 0.5f);
    };
    {
      // the reference pushes the table's edges in reverse (marching_cubes.h:88-93)
      const f3 p0 = edge_vertex((int)((row >> (4 * (3 * t + 2))) & 15));
      const f3 p1 = edge_vertex((int)((row >> (4 * (3 * t + 1))) & 15));
      const f3 p2 = edge_vertex((int)((row >> (4 * (3 * t))) & 15));
      const f3 n = f3_normalized(f3_cross(f3_sub(p1, p0), f3_sub(p2, p0)));
      float* v = d.verts + (size_t)tri * 9;
      float* nn = d.normals + (size_t)tri * 9;
      const f3 p[3] = {p0, p1, p2};
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        v[3 * j] = p[j].x; v[3 * j + 1] = p[j].y; v[3 * j + 2] = p[j].z;
        nn[3 * j] = n.x; nn[3 * j + 1] = n.y; nn[3 * j + 2] = n.z;
      }
      if (d.colors) {
        // updateMeshColor, mesh_integrator.h:372-392: nearest voxel of each vertex
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          // Block::computeVoxelIndexFromCoordinates, block.h:65-70
          const l3 vi = grid_index_from_point(f3_sub(p[j], origin), m.voxel_size_inv);
          uint32_t s = slot;
          int lx = (int)vi.x, ly = (int)vi.y, lz = (int)vi.z;
          if (lx < 0 || lx >= VPS || ly < 0 || ly >= VPS || lz < 0 || lz >= VPS) {
            // Layer::getBlockPtrByCoordinates + Block::getVoxelByCoordinates (layer.h:105-131,
            // block_inl.h:30-41: index relative to THAT block's origin, truncated into it)
            const l3 bi = grid_index_from_point(p[j], d.block_size_inv);
            const int ox = (int)bi.x - bx, oy = (int)bi.y - by, oz = (int)bi.z - bz;
            if ((unsigned)ox < 2u && (unsigned)oy < 2u && (unsigned)oz < 2u) {
              s = nslot[ox | (oy << 1) | (oz << 2)];
            } else {
              s = map_find(m, pack_block_key((int)bi.x, (int)bi.y, (int)bi.z));
              if (s != kInvalidSlot && !(m.blk_flags[s] & kFlagPublished)) s = kInvalidSlot;
            }
            const f3 no = {(float)(int)bi.x * d.block_size, (float)(int)bi.y * d.block_size,
                           (float)(int)bi.z * d.block_size};
            const l3 ti = grid_index_from_point(f3_sub(p[j], no), m.voxel_size_inv);
            lx = max(min((int)ti.x, VPS - 1), 0);
            ly = max(min((int)ti.y, VPS - 1), 0);
            lz = max(min((int)ti.z, VPS - 1), 0);
          }
          uint32_t col = 0;  // Color(): mesh->colors.resize() default
          if (s != kInvalidSlot) {
            const size_t a = (size_t)s * NV + (size_t)(lx + VPS * (ly + VPS * lz));
            if (!(m.weight[a] <= d.min_weight)) col = m.rgba[a];  // utils::getColorIfValid, meshing_utils.h:43-52
          }
          d.colors[(size_t)tri * 3 + j] = col;
        }
      }
    }
    }  // work items of this round
    __syncthreads();  // the queue is refilled by the next round
  }
}

}  // namespace
