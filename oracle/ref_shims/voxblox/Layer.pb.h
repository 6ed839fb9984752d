// DEPENDENCY SHIM (oracle/_ref build only): stand-in for the protoc-generated header of
// voxblox/proto/voxblox/Layer.proto.
#pragma once
#include <cstdint>
#include <string>
#include <google/protobuf/message.h>
namespace voxblox {
class LayerProto : public google::protobuf::Message {
 public:
  double voxel_size() const { return vs_; } void set_voxel_size(double v) { vs_ = v; }
  uint32_t voxels_per_side() const { return vps_; } void set_voxels_per_side(uint32_t v) { vps_ = v; }
  const std::string& type() const { return type_; } void set_type(const std::string& t) { type_ = t; }
 private:
  double vs_ = 0; uint32_t vps_ = 0; std::string type_;
};
}  // namespace voxblox
