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

/// What the mirror last left in (or last took from) one host block: the Block object itself — a shared reference, so that
/// the object stays readable even after the Layer dropped it and `use_count() == 1` says exactly that (removeBlock,
/// removeDistantBlocks; kReplace / removeBlock + re-allocate put a NEW object under the same BlockIndex,
/// layer_inl.h:203-210) —, its Update bits (consumers only ever clear bits: mesh_integrator.h:181,
/// esdf_integrator.cc:116-118; bits that appear were set by the host — Block::mergeBlock block_inl.h:120,
/// Layer::addBlockFromProto layer_inl.h:227) and two fingerprints of the voxel array (in-place writes without any marker:
/// deserializeMsgToLayer's kUpdate, conversions_inl.h:80-88): every 64-byte line, and kSampledLines spread lines.
template <typename VoxelType>
struct HostBlockRecord {
  BlockIndex index = BlockIndex::Zero();
  typename Block<VoxelType>::Ptr block;
  uint8_t bits = 0;
  uint64_t fingerprint = 0;   // every line
  uint64_t sampled = 0;       // kSampledLines lines
};
constexpr int kSampledLines = 8;

/// The blocks the device holds, as the host last saw them: a dense array (the per-call scan walks it with prefetches, on
/// helper threads when it is long) + BlockIndex -> position.
template <typename VoxelType>
struct KnownBlocks {
  typedef HostBlockRecord<VoxelType> Rec;
  std::vector<Rec> recs;
  typename AnyIndexHashMapType<size_t>::type pos;   // block_hash.h:33-41
  size_t cursor = 0;                                // rotating window of the touched-only reconcile
  size_t size() const { return recs.size(); }
  bool empty() const { return recs.empty(); }
  void clear() { recs.clear(); pos.clear(); cursor = 0; }
  Rec* find(const BlockIndex& bi) {
    auto it = pos.find(bi);
    return it == pos.end() ? nullptr : &recs[it->second];
  }
  Rec& operator[](const BlockIndex& bi) {
    auto it = pos.find(bi);
    if (it != pos.end()) return recs[it->second];
    pos.emplace(bi, recs.size());
    recs.emplace_back();
    recs.back().index = bi;
    return recs.back();
  }
  void erase(size_t i) {   // swap with the last
    pos.erase(recs[i].index);
    if (i + 1 != recs.size()) {
      recs[i] = std::move(recs.back());
      pos[recs[i].index] = i;
    }
    recs.pop_back();
  }
};

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
  KnownBlocks<TsdfVoxel> tsdf_known;             // the blocks the device holds, as the host last saw them
  KnownBlocks<EsdfVoxel> esdf_known;
  bool tsdf_check_all = false, esdf_check_all = false;   // markLayerEdited: the next reconcile fingerprints every line of every block
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
/// (vbx_blocks_remove), blocks that are new, replaced, carry Update bits the mirror did not leave, or whose voxel
/// fingerprint moved are uploaded (vbx_blocks_upload) — one batched call each.
///
/// What a call looks at (reconcileMode()):
///   -1 (default) O(touched), the reference's own cost model (its integrators only ever visit the blocks a cloud
///      touches): a pass over the mirror's OWN records — is the Block object still the Layer's (use_count), did Update
///      bits appear — which finds removals, replacements, merges and loads without a hash lookup or a voxel read, plus a
///      rotating window of kWindowBlocks blocks per call that are looked up in the Layer and compared by their sampled
///      fingerprint (so small maps are checked whole every call and big ones in turn), plus a membership scan when the
///      Layer's block count differs from the mirror's.  An IN-PLACE overwrite without any marker outside the window is the
///      one edit this mode does not see at once: announce it (markLayerEdited / markBlockEdited), or run mode 0;
///    0 every 64-byte line of every host block on every call (round 5's default; O(map): 12 MB per 251 blocks);
///   >0 the sampled fingerprint of every host block on every call (rounds 3-4).
/// VBX_DROPIN_FINGERPRINT_LINES in the environment sets the start value (unset = -1), vbx_dropin_set_reconcile_mode(int)
/// changes it.
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

/// The reconcile mode (see above), process-wide.
std::atomic<int>& reconcileMode();
constexpr size_t kWindowBlocks = 256;
/// The caller wrote voxels of `layer` in place without touching Update bits (deserializeMsgToLayer's kUpdate): the next
/// drop-in call on the layer compares every line of every block (markLayerEdited) / uploads the named block
/// (markBlockEdited).  Not needed for removeBlock / removeDistantBlocks / removeAllBlocks / LoadBlocksFromFile /
/// mergeBlock / addBlockFromProto — those change the Block object, the block count or the Update bits — nor in mode 0.
/// Also exported as extern "C" vbx_dropin_mark_layer_edited(const void* tsdf_or_esdf_layer).
void markLayerEdited(const Layer<TsdfVoxel>* tsdf_layer);
void markLayerEdited(const Layer<EsdfVoxel>* esdf_layer);
void markBlockEdited(const Layer<TsdfVoxel>* tsdf_layer, const BlockIndex& block_index);

/// Fingerprint of a block's voxel array over `lines` evenly spread 64-byte lines (0: EVERY line — a single-voxel poke
/// anywhere in the block moves it: each line is folded with odd multipliers — a change of one word changes the line's
/// value — and passed through a bijective finaliser before the lines are summed, so lines can be hashed independently
/// and in any order).
uint64_t voxelFingerprint(const void* voxels, size_t bytes, int lines = 0);

/// f(0) .. f(n-1) on the calling thread plus a few persistent helpers (VBX_DROPIN_THREADS, default min(8, hardware));
/// returns when all are done.  The per-call passes over whole blocks (fingerprints, voxel copies) are memory-bound.
void parallelFor(size_t n, const std::function<void(size_t)>& f);


}  // namespace hip
}  // namespace voxblox

#endif  // VOXBLOX_HIP_DEVICE_MIRROR_H_
