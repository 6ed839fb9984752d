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

#include <atomic>
#include <cstdint>
#include <functional>
#include <vector>

#include <vbx_hip.h>

#include "voxblox/core/block_hash.h"
#include "voxblox/core/layer.h"
#include "voxblox/core/voxel.h"

namespace voxblox {
namespace hip {

/// What the mirror last left in (or last took from) one host block: the Block object's address (kReplace /
/// removeBlock + re-allocate put a NEW object under the same BlockIndex, layer_inl.h:203-210), its Update bits
/// (consumers only ever clear bits: mesh_integrator.h:181, esdf_integrator.cc:116-118; bits that appear were set
/// by the host — Block::mergeBlock block_inl.h:120, Layer::addBlockFromProto layer_inl.h:227) and a sampled
/// fingerprint of the voxel array (in-place writes without any marker: deserializeMsgToLayer's kUpdate,
/// conversions_inl.h:80-88).
struct HostBlockRecord {
  const void* block = nullptr;
  uint8_t bits = 0;
  uint64_t fingerprint = 0;
};
typedef AnyIndexHashMapType<HostBlockRecord>::type HostBlockRecords;  // block_hash.h:33-41

/// Page-locked staging for the device-to-host copies of the mirror (grows, never shrinks; plain memory if pinning fails).
struct PinnedStaging {
  void* p = nullptr;
  size_t cap = 0;
  bool pinned = false;
  void* ensure(size_t bytes);
  ~PinnedStaging();
};

struct DeviceMirror {
  vbx_ctx* ctx = nullptr;
  const Layer<EsdfVoxel>* esdf_layer = nullptr;  // set by the EsdfIntegrator that shares the map
  bool esdf_pending = false;                     // addNewRobotPosition since the last update
  bool esdf_pending_ordered = false;             // ... queued in the reference's order (vbx_esdf_cfg::reference_order of that call)
  uint64_t last_use = 0;                         // LRU stamp of the association table
  std::atomic<int> pins{0};                      // drop-in calls in flight on this mirror (MirrorRef); never evicted while > 0
  uint64_t frames_integrated = 0;                // > 0: the device holds integrator state the host layer does not
                                                 // (FastTsdfIntegrator's approximate sets and frame counter)
  HostBlockRecords tsdf_known, esdf_known;       // the blocks the device holds, as the host last saw them
  std::vector<TsdfVoxel> tsdf_staging;           // host -> device (reconcile)
  std::vector<EsdfVoxel> esdf_staging;
  PinnedStaging down_staging;                    // device -> host (mirror)
  std::vector<int32_t> idx;
  std::vector<int32_t> new_idx;   // vbx_blocks_new_ordered: the blocks a call added, in the reference's insertion sequence
  std::vector<uint8_t> bits, has_data;
  // reconcile statistics (tests, INTEGRATION.md figures)
  uint64_t uploaded_blocks = 0, removed_blocks = 0;
};

/// A pinned reference to a mirror: while one exists the association table will not drop the mirror (a call on
/// another thread that creates the ninth live mirror would otherwise destroy a map this call is still using).
class MirrorRef {
 public:
  explicit MirrorRef(DeviceMirror* m) : m_(m) { m_->pins.fetch_add(1, std::memory_order_relaxed); }
  MirrorRef(MirrorRef&& o) noexcept : m_(o.m_) { o.m_ = nullptr; }
  MirrorRef(const MirrorRef&) = delete;
  MirrorRef& operator=(const MirrorRef&) = delete;
  ~MirrorRef() { if (m_) m_->pins.fetch_sub(1, std::memory_order_release); }
  DeviceMirror& operator*() const { return *m_; }
  DeviceMirror* operator->() const { return m_; }

 private:
  DeviceMirror* m_;
};

/// The device map of a host TSDF layer (created on first use), pinned for the lifetime of the returned reference.
/// The association table is keyed by the layer's address and bounded: beyond kMaxMirrors live entries the least
/// recently used one that is neither pinned nor waiting for an ESDF update is dropped (mirrors without integrator
/// state of their own first) — safe, because the host layer is coherent after every call and a layer that comes back
/// is uploaded again by reconcile*().  (If every entry is exempt the table simply grows.)
MirrorRef mirrorOf(Layer<TsdfVoxel>* tsdf_layer);
/// Drops the association and frees the device map (a Layer has no destructor hook the drop-in could use).
void releaseMirror(const Layer<TsdfVoxel>* tsdf_layer);

/// HOST -> DEVICE, on entry to every drop-in call.  The host Layer is the source of truth between calls: its
/// callers edit it directly (Layer::removeDistantBlocks tsdf_server.cc:315, removeAllBlocks, io::LoadBlocksFromFile
/// :566-578, deserializeMsgToLayer :639-653).  Blocks the host no longer has are removed on the device
/// (vbx_blocks_remove), blocks that are new, replaced, carry Update bits the mirror did not leave, or whose
/// sampled voxel fingerprint moved are uploaded (vbx_blocks_upload) — one batched call each.
void reconcileTsdfFromHost(DeviceMirror& dev, Layer<TsdfVoxel>* tsdf_layer);
void reconcileEsdfFromHost(DeviceMirror& dev, Layer<EsdfVoxel>* esdf_layer);

/// DEVICE -> HOST, on return: copies every TSDF block carrying the kMap bit on the device into the host layer
/// (AoS voxels, updated bits, has_data) and clears the device's kMap bits (they double as the mirror's dirty set).
void mirrorTsdfToHost(DeviceMirror& dev, Layer<TsdfVoxel>* tsdf_layer);
/// The same for ESDF blocks.
void mirrorEsdfToHost(DeviceMirror& dev, Layer<EsdfVoxel>* esdf_layer);

/// Reconcile counters of a layer's mirror (0 if none): blocks uploaded to / removed from the device so far.
void mirrorStats(const Layer<TsdfVoxel>* tsdf_layer, uint64_t* uploaded_blocks, uint64_t* removed_blocks);

/// 1 (default): EsdfIntegrator computes the reference's own result (queue order replayed on the device); 0: the order-free
/// fixed point.  Process-wide; also exported from the drop-in as extern "C" vbx_dropin_set_esdf_reference_order(int).
std::atomic<int>& esdfReferenceOrder();

/// Fingerprint of a block's voxel array.  Default: EVERY 64-byte line (a single-voxel poke anywhere in the block moves it:
/// each line is folded with odd multipliers — a change of one word changes the line's value — and passed through a
/// bijective finaliser before the lines are summed, so lines can be hashed independently and in any order).
/// VBX_DROPIN_FINGERPRINT_LINES = n > 0 samples n evenly spread lines instead (cheaper, blind between the samples).
uint64_t voxelFingerprint(const void* voxels, size_t bytes);

/// f(0) .. f(n-1) on the calling thread plus a few persistent helpers (VBX_DROPIN_THREADS, default min(8, hardware));
/// returns when all are done.  The per-call passes over whole blocks (fingerprints, voxel copies) are memory-bound.
void parallelFor(size_t n, const std::function<void(size_t)>& f);


}  // namespace hip
}  // namespace voxblox

#endif  // VOXBLOX_HIP_DEVICE_MIRROR_H_
