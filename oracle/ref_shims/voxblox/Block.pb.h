// DEPENDENCY SHIM (oracle/_ref build only): stand-in for the protoc-generated header of
// voxblox/proto/voxblox/Block.proto (field names from that file).
#pragma once
#include <cstdint>
#include <vector>
#include <google/protobuf/message.h>
namespace voxblox {
class BlockProto : public google::protobuf::Message {
 public:
  int32_t voxels_per_side() const { return vps_; } void set_voxels_per_side(int32_t v) { vps_ = v; }
  double voxel_size() const { return vs_; } void set_voxel_size(double v) { vs_ = v; }
  double origin_x() const { return ox_; } void set_origin_x(double v) { ox_ = v; }
  double origin_y() const { return oy_; } void set_origin_y(double v) { oy_ = v; }
  double origin_z() const { return oz_; } void set_origin_z(double v) { oz_ = v; }
  bool has_data() const { return hd_; } void set_has_data(bool v) { hd_ = v; }
  const std::vector<uint32_t>& voxel_data() const { return data_; }
  int voxel_data_size() const { return static_cast<int>(data_.size()); }
  void add_voxel_data(uint32_t w) { data_.push_back(w); }
 private:
  int32_t vps_ = 0; double vs_ = 0, ox_ = 0, oy_ = 0, oz_ = 0; bool hd_ = false; std::vector<uint32_t> data_;
};
}  // namespace voxblox
