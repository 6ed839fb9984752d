// vbx_sharded.hpp — C++ host class over libvbx_shard.so (include/vbx_shard.h): the multi-GPU ray-bundle
// sharding a voxblox_ros node links, one process per GPU.  Header-only, same conventions as
// vbx_integrators.hpp (abort-on-error like glog CHECK).
//
//   DeviceMap persistent(voxel, vps), delta(voxel, vps);
//   ShardedTsdfIntegrator it(kind, config, &persistent, &delta, rank, world, comm_id /* from rank 0 */);
//   per time step:  it.beginStep();  it.integratePointCloudDevice(T_G_C, d_points, d_rgba, n) ...;  it.endStep();
//
// integratePointCloudDevice takes device-resident points / colours (the sensor driver's DMA target); the
// persistent map of every rank holds the blocks that rank owns (vbx_shard_owner_of).
#pragma once

#include <cstdio>
#include <cstdlib>

#include "../../include/vbx_shard.h"
#include "vbx_integrators.hpp"

namespace vbx_host {

class ShardedTsdfIntegrator {
 public:
  ShardedTsdfIntegrator(TsdfIntegratorType type, const TsdfIntegratorBase::Config& config, DeviceMap* persistent,
                        DeviceMap* delta, int rank, int world, const uint8_t* comm_id, int device = 0)
      : kind_(static_cast<int>(type)), config_(config) {
    shard_ = vbx_shard_create(persistent->ctx(), delta->ctx(), rank, world, comm_id, device);
    if (!shard_) die(vbx_shard_last_error(nullptr));
  }
  ~ShardedTsdfIntegrator() { vbx_shard_destroy(shard_); }
  ShardedTsdfIntegrator(const ShardedTsdfIntegrator&) = delete;
  ShardedTsdfIntegrator& operator=(const ShardedTsdfIntegrator&) = delete;

  /// rank 0: the id every rank passes to the constructor
  static void createCommId(uint8_t id[VBX_SHARD_ID_BYTES]) {
    if (vbx_shard_get_unique_id(id) != VBX_OK) die("ncclGetUniqueId failed");
  }
  /// further delta maps: shard i of a step goes into delta i % n, concurrently (vbx_shard.h)
  void addDelta(DeviceMap* delta) { check(vbx_shard_add_delta(shard_, delta->ctx())); }
  /// two alternating sets of delta maps: endStep() returns while the exchange runs behind the next step
  void setPipelined(bool on) { check(vbx_shard_set_pipelined(shard_, on ? 1 : 0)); }
  void wait() { check(vbx_shard_wait(shard_)); }
  void beginStep() { check(vbx_shard_begin_step(shard_)); }
  void integratePointCloudDevice(const Transformation& T_G_C, const float* d_points_C, const uint8_t* d_rgba, size_t n,
                                 bool freespace_points = false) {
    const vbx_tsdf_cfg cfg = TsdfIntegratorBase::toC(config_);
    check(vbx_shard_integrate(shard_, kind_, &cfg, &T_G_C.getPosition().x, T_G_C.getRotationWxyz().data(), d_points_C, d_rgba, n,
                              freespace_points ? 1 : 0));
  }
  /// collective: every rank calls it once per step
  void endStep(bool apply_caps = false) {
    check(vbx_shard_end_step(shard_, apply_caps ? 1 : 0, config_.default_truncation_distance, config_.max_weight));
  }
  vbx_shard_stats stats() const {
    vbx_shard_stats s{};
    vbx_shard_get_stats(shard_, &s);
    return s;
  }

 private:
  static void die(const char* msg) {
    std::fprintf(stderr, "voxblox (HIP, sharded): %s\n", msg);
    std::abort();
  }
  void check(int rc) {
    if (rc != VBX_OK) die(vbx_shard_last_error(shard_));
  }
  vbx_shard* shard_ = nullptr;
  int kind_;
  TsdfIntegratorBase::Config config_;
};

}  // namespace vbx_host
