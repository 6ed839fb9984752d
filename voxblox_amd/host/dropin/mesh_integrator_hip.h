// mesh_integrator_hip.h — MeshIntegrator<TsdfVoxel>::generateMesh on the MI355X path.
//
// voxblox's mesher is a header-only class template (include/voxblox/mesh/mesh_integrator.h, unchanged), so its drop-in is
// an explicit specialisation of the ONE member that does the work,
//   MeshIntegrator<TsdfVoxel>::generateMesh(only_mesh_updated_blocks, clear_updated_flag)      mesh_integrator.h:142-172
// (per block updateMeshForBlock :250-270 = extractBlockMesh :197-248 + MarchingCubes::meshCube marching_cubes.h:70-111 +
// updateMeshColor :372-392).  Everything else — constructors, Config, MeshLayer, Mesh, every consumer of the MeshLayer
// (voxblox_ros/src/tsdf_server.cc:420-470, mesh_vis.h, mesh_ply.h, mesh_utils.h) — is the reference's own code.
// MeshIntegrator<EsdfVoxel> and subclasses that override updateMeshForBlock keep the reference's CPU path.
//
// How it gets in: include this header wherever voxblox/mesh/mesh_integrator.h is included before MeshIntegrator<TsdfVoxel>
// is first used — in practice ONE line at the end of mesh_integrator.h (`#include "mesh_integrator_hip.h"`), or
// `-include mesh_integrator_hip.h` on the compile line of the translation units that mesh (tsdf_server.cc) — and add
// mesh_integrator_hip.cc to libvoxblox next to tsdf_integrator_hip.cc / esdf_integrator_hip.cc.  The specialisation must
// be DECLARED before the first implicit instantiation; that is all this header does.
#ifndef VOXBLOX_HIP_MESH_INTEGRATOR_HIP_H_
#define VOXBLOX_HIP_MESH_INTEGRATOR_HIP_H_

#include "voxblox/mesh/mesh_integrator.h"

namespace voxblox {
namespace hip {
/// The body of the specialisation (mesh_integrator_hip.cc): the blocks the reference would mesh — the HOST layer's
/// getAllUpdatedBlocks(Update::kMesh) / getAllAllocatedBlocks, in the host layer's iteration order, so that the MeshLayer's
/// own container is filled in the reference's sequence — are meshed on the device from the layer's HBM copy
/// (vbx_mesh_generate) and their vertices / normals / colours / indices written into the caller's MeshLayer.
void generateMeshOnDevice(const MeshIntegratorConfig& config, const Layer<TsdfVoxel>* sdf_layer_const,
                          Layer<TsdfVoxel>* sdf_layer_mutable, MeshLayer* mesh_layer, bool only_mesh_updated_blocks,
                          bool clear_updated_flag);
}  // namespace hip

template <>
inline void MeshIntegrator<TsdfVoxel>::generateMesh(bool only_mesh_updated_blocks, bool clear_updated_flag) {
  CHECK(!clear_updated_flag || (sdf_layer_mutable_ != nullptr))                                    // :143-146
      << "If you would like to modify the updated flag in the blocks, please "
      << "use the constructor that provides a non-const link to the sdf "
      << "layer!";
  hip::generateMeshOnDevice(config_, sdf_layer_const_, sdf_layer_mutable_, mesh_layer_, only_mesh_updated_blocks,
                            clear_updated_flag);
}

}  // namespace voxblox

#endif  // VOXBLOX_HIP_MESH_INTEGRATOR_HIP_H_
