// device_mirror.h — drop-in glue shared by tsdf_integrator_hip.cc and esdf_integrator_hip.cc.
//
// These three files are what a voxblox maintainer adds to libvoxblox INSTEAD OF
// src/integrator/tsdf_integrator.cc and src/integrator/esdf_integrator.cc: they are compiled against
// voxblox's own, unchanged headers (include/voxblox/integrator/{tsdf,esdf}_integrator.h) and define the
// same symbols, so voxblox_ros, the tests and the planners link and run unchanged (SURVEY 8(b):
// link-time substitution — TsdfServer instantiates the concrete classes itself, tsdf_server.cc:92-106).
// The only dependency is the C-ABI of include/vbx_hip.h.
//
// One HBM-resident map (vbx_ctx) per host Layer<TsdfVoxel>; the ESDF layer of the same map shares it
// ("block indices are the same across all layers", esdf_integrator.cc:144).  The integrator classes
// cannot grow members (their headers are the reference's), so the association lives in a table keyed
// by the TSDF layer's address.
#ifndef VOXBLOX_HIP_DEVICE_MIRROR_H_
#define VOXBLOX_HIP_DEVICE_MIRROR_H_

#include <cstdint>
#include <vector>

#include <vbx_hip.h>

#include "voxblox/core/layer.h"
#include "voxblox/core/voxel.h"

namespace voxblox {
namespace hip {

struct DeviceMirror {
  vbx_ctx* ctx = nullptr;
  const Layer<EsdfVoxel>* esdf_layer = nullptr;  // set by the EsdfIntegrator that shares the map
  bool esdf_pending = false;                     // addNewRobotPosition since the last update
  std::vector<TsdfVoxel> tsdf_staging;
  std::vector<EsdfVoxel> esdf_staging;
  std::vector<int32_t> idx;
  std::vector<uint8_t> bits, has_data;
};

/// The device map of a host TSDF layer (created on first use; a host layer without blocks resets it:
/// a fresh Layer at a recycled address, or removeAllBlocks()).
DeviceMirror& mirrorOf(Layer<TsdfVoxel>* tsdf_layer);
/// Drops the association (call before destroying a Layer whose address may be reused while blocks remain).
void releaseMirror(const Layer<TsdfVoxel>* tsdf_layer);

/// Copies every TSDF block carrying the kMap bit on the device into the host layer (AoS voxels,
/// updated bits, has_data) and clears the device's kMap bits (they double as the mirror's dirty set).
void mirrorTsdfToHost(DeviceMirror& dev, Layer<TsdfVoxel>* tsdf_layer);
/// The same for ESDF blocks.
void mirrorEsdfToHost(DeviceMirror& dev, Layer<EsdfVoxel>* esdf_layer);

}  // namespace hip
}  // namespace voxblox

#endif  // VOXBLOX_HIP_DEVICE_MIRROR_H_
