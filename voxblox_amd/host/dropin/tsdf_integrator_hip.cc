// tsdf_integrator_hip.cc — replaces voxblox/src/integrator/tsdf_integrator.cc.
//
// Same classes, same signatures (include/voxblox/integrator/tsdf_integrator.h, unchanged):
//   TsdfIntegratorFactory::create                  tsdf_integrator.cc:8-46
//   TsdfIntegratorBase::TsdfIntegratorBase/setLayer tsdf_integrator.cc:53-80
//   Simple/Merged/FastTsdfIntegrator::integratePointCloud   :242-305, :307-338, :555-590
// The bodies hand the cloud to the MI355X path through the C-ABI (vbx_tsdf_integrate) and then copy
// the blocks the call touched back into the caller's host Layer, so everything that reads the Layer
// between calls (mesher, publishers, save_map, ESDF on the CPU) keeps working.  Failure = glog CHECK
// (abort), the reference's own error convention.
#include "voxblox/integrator/tsdf_integrator.h"

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <sstream>
#include <thread>

#include "device_mirror.h"
#include "voxblox/utils/timing.h"

namespace voxblox {
namespace hip {
namespace {
std::mutex g_mu;
std::map<const void*, DeviceMirror*>& table() {
  static std::map<const void*, DeviceMirror*> t;
  return t;
}
constexpr size_t kMaxMirrors = 8;       // live (layer address -> device map) associations before the LRU one goes
constexpr uint32_t kInitialBlocks = 4096;  // initial pool of a mirror (~200 MB at vps 16); it doubles on demand
uint64_t g_tick = 0;

void destroyMirror(DeviceMirror* m) {
  if (m->ctx) vbx_destroy(m->ctx);
  delete m;
}

// (No cleanup at process exit: the HIP runtime may already be gone when static destructors run; the driver frees
// a dying process's device memory.  Long-lived processes drop maps through releaseMirror() or the LRU bound.)

// ---- a handful of persistent helper threads for the per-call passes over whole blocks ---------------------------------
class Helpers {
 public:
  static Helpers& get() {
    static Helpers* h = new Helpers;   // (never destroyed: joining threads from a static destructor is asking for trouble)
    return *h;
  }
  void run(size_t n, const std::function<void(size_t)>& f) {
    if (n == 0) return;
    if (threads_.empty() || n < 4) {
      for (size_t i = 0; i < n; ++i) f(i);
      return;
    }
    std::unique_lock<std::mutex> call_lock(call_mu_);   // one parallelFor at a time
    {
      std::lock_guard<std::mutex> lock(mu_);
      fn_ = &f;
      n_ = n;
      next_.store(0, std::memory_order_relaxed);
      busy_ = threads_.size();
      ++generation_;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> lock(mu_);
    done_cv_.wait(lock, [this] { return busy_ == 0; });
    fn_ = nullptr;
  }

 private:
  Helpers() {
    const char* e = getenv("VBX_DROPIN_THREADS");
    const unsigned hw = std::thread::hardware_concurrency();
    int want = e ? atoi(e) : static_cast<int>(std::min(8u, hw ? hw : 1u));
    for (int i = 1; i < want; ++i) threads_.emplace_back([this] { loop(); });
    for (std::thread& t : threads_) t.detach();
  }
  void work() {
    for (;;) {
      const size_t i = next_.fetch_add(1, std::memory_order_relaxed);
      if (i >= n_) break;
      (*fn_)(i);
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lock(mu_);
        cv_.wait(lock, [&] { return generation_ != seen; });
        seen = generation_;
      }
      work();
      {
        std::lock_guard<std::mutex> lock(mu_);
        --busy_;
      }
      done_cv_.notify_one();
    }
  }
  std::vector<std::thread> threads_;
  std::mutex mu_, call_mu_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(size_t)>* fn_ = nullptr;
  size_t n_ = 0, busy_ = 0;
  std::atomic<size_t> next_{0};
  uint64_t generation_ = 0;
};
}  // namespace

void parallelFor(size_t n, const std::function<void(size_t)>& f) { Helpers::get().run(n, f); }

void* PinnedStaging::ensure(size_t bytes) {
  if (bytes <= cap) return p;
  if (p) {
    if (pinned) vbx_host_free(p); else free(p);
  }
  cap = bytes + bytes / 4;
  p = vbx_host_alloc(cap);
  pinned = p != nullptr;
  if (!p) p = malloc(cap);
  CHECK(p != nullptr) << "out of host memory for the mirror's staging buffer";
  return p;
}
PinnedStaging::~PinnedStaging() {
  if (!p) return;
  if (pinned) vbx_host_free(p); else free(p);
}

std::atomic<int>& reconcileMode() {
  static std::atomic<int> v([] {
    const char* e = getenv("VBX_DROPIN_FINGERPRINT_LINES");
    return (e && e[0]) ? atoi(e) : -1;   // unset: the O(touched) reconcile (round 6); 0: every line of every block (round 5); n: sampled (rounds 3-4)
  }());
  return v;
}

uint64_t voxelFingerprint(const void* voxels, size_t bytes, int want) {
  // Per 64-byte line: the eight 8-byte words folded with odd multipliers (a change of any one word changes the sum), the
  // line number mixed in, a bijective finaliser (murmur3's fmix64) on top; the block's fingerprint is the SUM of its lines'
  // values — independent per line, so the pass is eight multiply chains wide and memory-bound.
  static const uint64_t K[8] = {0x9E3779B97F4A7C15ull, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull, 0xD6E8FEB86659FD93ull,
                                0xFF51AFD7ED558CCDull, 0xC4CEB9FE1A85EC53ull, 0x2545F4914F6CDD1Dull, 0x9FB21C651E98DF25ull};
  const size_t n_lines = bytes / 64;
  const size_t lines = (want <= 0 || (size_t)want > n_lines) ? n_lines : (size_t)want;
  const unsigned char* base = static_cast<const unsigned char*>(voxels);
  uint64_t h = 0x9E3779B97F4A7C15ull ^ bytes;
  for (size_t k = 0; k < lines; ++k) {
    const size_t line = (lines == n_lines) ? k : (k * n_lines) / lines + (n_lines / lines) / 2;
    uint64_t w[8];
    std::memcpy(w, base + line * 64, 64);
    uint64_t v = line * 0xD1B54A32D192ED03ull;
    for (int j = 0; j < 8; ++j) v += (w[j] + j + 1) * K[j];
    v ^= v >> 33;
    v *= 0xFF51AFD7ED558CCDull;
    v ^= v >> 33;
    v *= 0xC4CEB9FE1A85EC53ull;
    v ^= v >> 33;
    h += v;
  }
  // (the tail beyond the last whole line: 12 B and 20 B voxels times 4096 are multiples of 64)
  return h;
}

MirrorRef mirrorOf(Layer<TsdfVoxel>* layer) {
  CHECK_NOTNULL(layer);
  std::lock_guard<std::mutex> lock(g_mu);
  DeviceMirror*& m = table()[layer];
  if (m != nullptr) {  // a different Layer at a recycled address: the geometry tells
    vbx_map_cfg have;
    CHECK_EQ(vbx_get_map_cfg(m->ctx, &have), VBX_OK) << vbx_last_error(m->ctx);
    if (have.voxel_size != layer->voxel_size() || have.voxels_per_side != layer->voxels_per_side()) {
      CHECK_EQ(m->pins.load(), 0) << "a Layer was replaced at the same address while a drop-in call on it is running";
      destroyMirror(m);
      m = nullptr;
    }
  }
  if (m == nullptr) {
    m = new DeviceMirror;
    vbx_map_cfg cfg;
    cfg.voxel_size = layer->voxel_size();
    cfg.voxels_per_side = static_cast<uint32_t>(layer->voxels_per_side());
    cfg.max_blocks = kInitialBlocks;
    m->ctx = vbx_create(&cfg, /*device=*/0);
    CHECK(m->ctx != nullptr) << vbx_last_error(nullptr);
    if (table().size() > kMaxMirrors) {  // bound the table: drop the least recently used association
      // never a pinned one or one with an ESDF update pending; one without integrator state of its own before one with
      // (the Fast integrator's approximate sets and frame counter live on the device only: losing them is what the
      // reference's own periodic reset does, tsdf_integrator.cc:564-573, but it is not free)
      const void* victim = nullptr;
      uint64_t oldest = ~0ull;
      bool victim_stateful = true;
      for (auto& kv : table()) {
        const DeviceMirror* c = kv.second;
        if (kv.first == layer || !c || c->esdf_pending || c->pins.load(std::memory_order_acquire) != 0) continue;
        const bool stateful = c->frames_integrated != 0;
        if ((victim_stateful && !stateful) || (stateful == victim_stateful && c->last_use < oldest)) {
          oldest = c->last_use;
          victim = kv.first;
          victim_stateful = stateful;
        }
      }
      if (victim) {
        destroyMirror(table()[victim]);
        table().erase(victim);
      }
    }
  }
  DeviceMirror* dev = table()[layer];
  dev->last_use = ++g_tick;
  return MirrorRef(dev);   // (pinned under g_mu: an eviction on another thread sees the pin)
}

void releaseMirror(const Layer<TsdfVoxel>* layer) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = table().find(layer);
  if (it == table().end()) return;
  CHECK_EQ(it->second->pins.load(), 0) << "releaseMirror() while a drop-in call on the layer is running";
  destroyMirror(it->second);
  table().erase(it);
}

void mirrorStats(const Layer<TsdfVoxel>* layer, uint64_t* uploaded_blocks, uint64_t* removed_blocks) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = table().find(layer);
  if (uploaded_blocks) *uploaded_blocks = it == table().end() ? 0 : it->second->uploaded_blocks;
  if (removed_blocks) *removed_blocks = it == table().end() ? 0 : it->second->removed_blocks;
}

void markLayerEdited(const Layer<TsdfVoxel>* layer) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = table().find(layer);
  if (it != table().end()) it->second->tsdf_check_all = true;   // (no mirror yet: everything is uploaded at the first call anyway)
}
void markLayerEdited(const Layer<EsdfVoxel>* layer) {
  std::lock_guard<std::mutex> lock(g_mu);
  for (auto& kv : table())
    if (kv.second && kv.second->esdf_layer == layer) kv.second->esdf_check_all = true;
}
void markBlockEdited(const Layer<TsdfVoxel>* layer, const BlockIndex& block_index) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = table().find(layer);
  if (it == table().end()) return;
  // a fingerprint that cannot match: the block goes up at the next call's window / scan — the record is put where the
  // rotating window starts, so the very next reconcile looks at it
  KnownBlocks<TsdfVoxel>& known = it->second->tsdf_known;
  HostBlockRecord<TsdfVoxel>* rec = known.find(block_index);
  if (!rec) return;   // unknown to the mirror: found as a new block anyway
  rec->block.reset();   // "replaced": the scan's use_count test sends it through the upload path
}

namespace {
// Shared by the two layers: which host blocks must go up, which device blocks must go.
template <typename VoxelType>
void reconcileFromHost(DeviceMirror& dev, Layer<VoxelType>* layer, int vbx_layer, KnownBlocks<VoxelType>* known,
                       std::vector<VoxelType>* staging, bool* check_all) {
  typedef HostBlockRecord<VoxelType> Rec;
  const size_t nv = layer->voxels_per_side() * layer->voxels_per_side() * layer->voxels_per_side();
  const size_t bytes = nv * sizeof(VoxelType);
  const size_t host_n = layer->getNumberOfAllocatedBlocks();
  if (host_n == 0u) {  // removeAllBlocks(), or a new Layer at a recycled address
    if (!known->empty()) {
      CHECK_EQ(vbx_clear(dev.ctx, vbx_layer), VBX_OK) << vbx_last_error(dev.ctx);
      dev.removed_blocks += known->size();
      known->clear();
    }
    *check_all = false;
    return;
  }
  int mode = reconcileMode().load(std::memory_order_relaxed);
  if (*check_all) mode = 0;
  *check_all = false;
  std::vector<int32_t> gone;
  std::vector<typename Block<VoxelType>::Ptr> up;   // blocks to upload, with their indices in dev.idx
  std::vector<BlockIndex> up_idx;
  auto note_gone = [&](const BlockIndex& bi) {
    gone.push_back(bi.x());
    gone.push_back(bi.y());
    gone.push_back(bi.z());
  };
  if (mode < 0) {
    // ---- O(touched).  1. the mirror's own records: is the object still the Layer's, did Update bits appear?  No hash
    // lookup, no voxel read: one or two cache lines per block (the shared_ptr's control block sits in front of the Block it
    // was made with), prefetched ahead, on the helper threads when the map is large.
    std::vector<Rec>& recs = known->recs;
    const size_t n = recs.size();
    std::vector<uint8_t> suspect(n, 0);
    auto scan = [&](size_t lo, size_t hi) {
      constexpr size_t kAhead = 8;
      for (size_t i = lo; i < hi; ++i) {
        if (i + kAhead < hi && recs[i + kAhead].block) __builtin_prefetch(recs[i + kAhead].block.get());
        Rec& r = recs[i];
        if (!r.block || r.block.use_count() <= 1) { suspect[i] = 1; continue; }   // dropped or replaced (or markBlockEdited)
        const uint8_t bits = static_cast<uint8_t>(r.block->updated().to_ulong());
        if (bits & ~r.bits) suspect[i] = 1;   // the host set bits: merged / loaded into
        else r.bits = bits;                   // consumers cleared bits: remember, so that a later set() shows
      }
    };
    constexpr size_t kChunk = 2048;
    if (n <= 2 * kChunk) scan(0, n);
    else parallelFor((n + kChunk - 1) / kChunk, [&](size_t c) { scan(c * kChunk, std::min(n, (c + 1) * kChunk)); });
    // 2. a rotating window: looked up in the Layer (identity) and compared by the sampled fingerprint
    const size_t w = std::min(n, kWindowBlocks);
    std::vector<size_t> win(w);
    for (size_t k = 0; k < w; ++k) win[k] = (known->cursor + k) % n;
    known->cursor = n ? (known->cursor + w) % n : 0;
    std::vector<typename Block<VoxelType>::Ptr> win_block(w);
    for (size_t k = 0; k < w; ++k)
      if (!suspect[win[k]]) win_block[k] = layer->getBlockPtrByIndex(recs[win[k]].index);
    parallelFor(w, [&](size_t k) {
      const size_t i = win[k];
      if (suspect[i]) return;
      if (win_block[k].get() != recs[i].block.get() ||
          voxelFingerprint(&win_block[k]->getVoxelByLinearIndex(0), bytes, kSampledLines) != recs[i].sampled)
        suspect[i] = 1;
    });
    // 3. the suspects: what does the Layer hold under that index now?  (descending, so that erase()'s swap never moves an
    // unvisited suspect)
    for (size_t i = n; i-- > 0;) {
      if (!suspect[i]) continue;
      const BlockIndex bi = recs[i].index;
      typename Block<VoxelType>::Ptr now = layer->getBlockPtrByIndex(bi);
      if (!now) {
        note_gone(bi);
        known->erase(i);
      } else {
        up.push_back(now);
        up_idx.push_back(bi);
      }
    }
    // 4. blocks the mirror has never seen (created by the host: loadMap, tsdfMapCallback), or a removal the scan could not
    // see because somebody else still holds the object: the counts tell, and only then is the Layer walked
    if (known->size() != host_n) {
      BlockIndexList host_blocks;
      layer->getAllAllocatedBlocks(&host_blocks);
      for (const BlockIndex& bi : host_blocks) {
        if (known->find(bi)) continue;
        up.push_back(layer->getBlockPtrByIndex(bi));
        up_idx.push_back(bi);
        (*known)[bi];   // (filled in below)
      }
      if (known->size() != host_n) {
        for (size_t i = known->size(); i-- > 0;) {
          if (layer->hasBlock(known->recs[i].index)) continue;
          note_gone(known->recs[i].index);
          known->erase(i);
        }
      }
    }
  } else {
    // ---- every host block, every call: all lines (mode 0) or the sampled ones (mode > 0)
    BlockIndexList host_blocks;
    layer->getAllAllocatedBlocks(&host_blocks);
    // 1. blocks the host dropped (removeDistantBlocks, removeBlock)
    for (size_t i = known->size(); i-- > 0;) {
      if (layer->hasBlock(known->recs[i].index)) continue;
      note_gone(known->recs[i].index);
      known->erase(i);
    }
    // 2. blocks the host created, replaced or wrote to
    std::vector<typename Block<VoxelType>::Ptr> all(host_blocks.size());
    std::vector<uint64_t> all_fp(host_blocks.size());
    {
      size_t k = 0;
      for (const BlockIndex& bi : host_blocks) all[k++] = layer->getBlockPtrByIndex(bi);
    }
    const int lines = mode == 0 ? 0 : kSampledLines;
    parallelFor(all.size(), [&](size_t i) { all_fp[i] = voxelFingerprint(&all[i]->getVoxelByLinearIndex(0), bytes, lines); });
    size_t k_block = 0;
    for (const BlockIndex& bi : host_blocks) {
      typename Block<VoxelType>::Ptr& block = all[k_block];
      const uint64_t fp = all_fp[k_block++];
      const uint8_t bits = static_cast<uint8_t>(block->updated().to_ulong());
      Rec* rec = known->find(bi);
      if (rec && rec->block.get() == block.get() && (bits & ~rec->bits) == 0 && (mode == 0 ? rec->fingerprint : rec->sampled) == fp) {
        rec->bits = bits;  // consumers cleared bits: remember, so that a later set() shows
        continue;
      }
      up.push_back(block);
      up_idx.push_back(bi);
    }
  }
  if (!gone.empty()) {
    CHECK_EQ(vbx_blocks_remove(dev.ctx, vbx_layer, gone.data(), gone.size() / 3), VBX_OK) << vbx_last_error(dev.ctx);
    dev.removed_blocks += gone.size() / 3;
  }
  if (up.empty()) return;
  dev.idx.clear();
  dev.bits.clear();
  dev.has_data.clear();
  staging->resize(up.size() * nv);
  std::vector<uint64_t> fp_all(up.size()), fp_s(up.size());
  parallelFor(up.size(), [&](size_t i) {
    const void* src = &up[i]->getVoxelByLinearIndex(0);
    std::memcpy(static_cast<void*>(staging->data() + i * nv), src, bytes);
    fp_all[i] = voxelFingerprint(src, bytes, 0);
    fp_s[i] = voxelFingerprint(src, bytes, kSampledLines);
  });
  for (size_t i = 0; i < up.size(); ++i) {
    const BlockIndex& bi = up_idx[i];
    const uint8_t bits = static_cast<uint8_t>(up[i]->updated().to_ulong());
    dev.idx.push_back(bi.x());
    dev.idx.push_back(bi.y());
    dev.idx.push_back(bi.z());
    // kMap is the MIRROR's dirty bit on the device (a block uploaded from the host is not dirty); kMesh / kEsdf
    // travel: the device-side ESDF update must see a loaded block as updated (esdf_integrator.cc:104-110)
    dev.bits.push_back(static_cast<uint8_t>(bits & ~VBX_UPDATE_MAP));
    dev.has_data.push_back(up[i]->has_data() ? 1 : 0);
    Rec& rec = (*known)[bi];
    rec.block = up[i];
    rec.bits = bits;
    rec.fingerprint = fp_all[i];
    rec.sampled = fp_s[i];
  }
  CHECK_EQ(vbx_blocks_upload(dev.ctx, vbx_layer, dev.idx.data(), up.size(), staging->data(), dev.bits.data(),
                             dev.has_data.data()),
           VBX_OK)
      << vbx_last_error(dev.ctx);
  dev.uploaded_blocks += up.size();
}
}  // namespace

void reconcileTsdfFromHost(DeviceMirror& dev, Layer<TsdfVoxel>* layer) {
  static_assert(sizeof(TsdfVoxel) == 12, "TsdfVoxel is {float distance; float weight; Color color}");
  reconcileFromHost<TsdfVoxel>(dev, layer, VBX_LAYER_TSDF, &dev.tsdf_known, &dev.tsdf_staging, &dev.tsdf_check_all);
}

void reconcileEsdfFromHost(DeviceMirror& dev, Layer<EsdfVoxel>* layer) {
  static_assert(sizeof(EsdfVoxel) == 20, "EsdfVoxel is {float distance; bool observed, hallucinated, in_queue, fixed; Vector3i parent}");
  reconcileFromHost<EsdfVoxel>(dev, layer, VBX_LAYER_ESDF, &dev.esdf_known, &dev.esdf_staging, &dev.esdf_check_all);
}

void mirrorTsdfToHost(DeviceMirror& dev, Layer<TsdfVoxel>* layer) {
  static_assert(sizeof(TsdfVoxel) == 12, "TsdfVoxel is {float distance; float weight; Color color}");
  // (one call into a buffer that is almost always large enough: every vbx_blocks_updated reads the pool's slot table back —
  // a state read-back and two copies — and the ask-for-the-count-first protocol paid that twice per frame)
  size_t n = 0;
  if (dev.idx.size() < 3 * 1024) dev.idx.resize(3 * 1024);
  size_t cap = dev.idx.size() / 3;
  CHECK_EQ(vbx_blocks_updated(dev.ctx, VBX_LAYER_TSDF, VBX_UPDATE_MAP, dev.idx.data(), cap, &n), VBX_OK)
      << vbx_last_error(dev.ctx);
  if (n == 0) return;
  if (n > cap) {
    dev.idx.resize(3 * (n + n / 2));
    cap = dev.idx.size() / 3;
    CHECK_EQ(vbx_blocks_updated(dev.ctx, VBX_LAYER_TSDF, VBX_UPDATE_MAP, dev.idx.data(), cap, &n), VBX_OK)
        << vbx_last_error(dev.ctx);
    CHECK_LE(n, cap);
  }
  const size_t nv = layer->voxels_per_side() * layer->voxels_per_side() * layer->voxels_per_side();
  {
    // updateLayerWithStoredBlocks (tsdf_integrator.cc:137-147): the blocks the call added join the host Layer in the
    // sequence the reference's single-threaded integrator inserts them — temp_block_map_'s iteration order, replayed by
    // the library from the first touch of every new block — so that the Layer's own unordered_map, and with it
    // getAllUpdatedBlocks and the ESDF's walk (layer.h:194-203, esdf_integrator.cc:104-143), iterate like in a CPU run
    size_t n_new = 0;
    if (dev.new_idx.size() < 3 * 1024) dev.new_idx.resize(3 * 1024);
    CHECK_EQ(vbx_blocks_new_ordered(dev.ctx, dev.new_idx.data(), dev.new_idx.size() / 3, &n_new), VBX_OK) << vbx_last_error(dev.ctx);
    if (n_new > dev.new_idx.size() / 3) {
      dev.new_idx.resize(3 * n_new);
      CHECK_EQ(vbx_blocks_new_ordered(dev.ctx, dev.new_idx.data(), n_new, &n_new), VBX_OK) << vbx_last_error(dev.ctx);
    }
    if (n_new) {
      for (size_t i = 0; i < n_new; ++i)
        layer->allocateBlockPtrByIndex(BlockIndex(dev.new_idx[3 * i], dev.new_idx[3 * i + 1], dev.new_idx[3 * i + 2]));
    }
  }
  TsdfVoxel* staging = static_cast<TsdfVoxel*>(dev.down_staging.ensure(n * nv * sizeof(TsdfVoxel)));   // page-locked
  dev.bits.resize(n);
  dev.has_data.resize(n);
  std::vector<Block<TsdfVoxel>::Ptr> blocks(n);
  std::vector<uint64_t> fps(n), fps_s(n);
  for (size_t i = 0; i < n; ++i)   // (the Layer's container is not thread-safe: allocation stays on this thread)
    blocks[i] = layer->allocateBlockPtrByIndex(BlockIndex(dev.idx[3 * i], dev.idx[3 * i + 1], dev.idx[3 * i + 2]));
  auto copy_in = [&](size_t lo, size_t hi) {
    parallelFor(hi - lo, [&](size_t k) {
      const size_t i = lo + k;
      const TsdfVoxel* src = staging + i * nv;
      std::memcpy(static_cast<void*>(&blocks[i]->getVoxelByLinearIndex(0)), src, nv * sizeof(TsdfVoxel));
      fps[i] = voxelFingerprint(src, nv * sizeof(TsdfVoxel), 0);
      fps_s[i] = voxelFingerprint(src, nv * sizeof(TsdfVoxel), kSampledLines);
    });
  };
  // (measured and left out, round 6: the download in two halves with the first half's host copy on a side thread behind the
  // second half's device copy — the extra read-back and the thread cost more than the overlap gave: mirror 0.41 -> 0.48 ms)
  CHECK_EQ(vbx_blocks_download(dev.ctx, VBX_LAYER_TSDF, dev.idx.data(), n, staging, dev.bits.data(), dev.has_data.data()), VBX_OK)
      << vbx_last_error(dev.ctx);
  copy_in(0, n);
  for (size_t i = 0; i < n; ++i) {
    const BlockIndex bi(dev.idx[3 * i], dev.idx[3 * i + 1], dev.idx[3 * i + 2]);
    Block<TsdfVoxel>::Ptr& block = blocks[i];
    // block->updated().set() on every touched block (tsdf_integrator.cc:128); bits a host consumer has
    // cleared since (mesher: kMesh, ESDF: kEsdf) come back only if the device set them again
    block->updated() |= std::bitset<Update::kCount>(dev.bits[i]);
    block->has_data() = dev.has_data[i] != 0;  // the integrators never set it
    HostBlockRecord<TsdfVoxel>& rec = dev.tsdf_known[bi];
    rec.block = block;
    rec.bits = static_cast<uint8_t>(block->updated().to_ulong());
    rec.fingerprint = fps[i];
    rec.sampled = fps_s[i];
  }
  // kMap doubles as the mirror's dirty bit on the device; kMesh / kEsdf stay for the device-side
  // mesher / ESDF
  CHECK_EQ(vbx_clear_updated(dev.ctx, VBX_LAYER_TSDF, VBX_UPDATE_MAP), VBX_OK) << vbx_last_error(dev.ctx);
}

namespace {
vbx_tsdf_cfg toC(const TsdfIntegratorBase::Config& c) {
  vbx_tsdf_cfg o;
  vbx_tsdf_cfg_default(&o);
  o.default_truncation_distance = c.default_truncation_distance;
  o.max_weight = c.max_weight;
  o.voxel_carving_enabled = c.voxel_carving_enabled;
  o.min_ray_length_m = c.min_ray_length_m;
  o.max_ray_length_m = c.max_ray_length_m;
  o.use_const_weight = c.use_const_weight;
  o.allow_clear = c.allow_clear;
  o.use_weight_dropoff = c.use_weight_dropoff;
  o.use_sparsity_compensation_factor = c.use_sparsity_compensation_factor;
  o.sparsity_compensation_factor = c.sparsity_compensation_factor;
  o.integrator_threads = static_cast<int32_t>(c.integrator_threads);
  if (c.integration_order_mode == "sorted") {
    o.integration_order_mode = 1;
  } else {
    CHECK(c.integration_order_mode == "mixed") << "Unknown integration order mode: '" << c.integration_order_mode
                                               << "'!";  // integrator_utils.cc:12
    o.integration_order_mode = 0;
  }
  o.enable_anti_grazing = c.enable_anti_grazing;
  o.start_voxel_subsampling_factor = c.start_voxel_subsampling_factor;
  o.max_consecutive_ray_collisions = c.max_consecutive_ray_collisions;
  o.clear_checks_every_n_frames = c.clear_checks_every_n_frames;
  o.max_integration_time_s = c.max_integration_time_s;
  // merged_bundle_order / fast_observed_set stay 0: the reference's own semantics
  return o;
}

void integrateOnDevice(int kind, const TsdfIntegratorBase::Config& config, Layer<TsdfVoxel>* layer,
                       const Transformation& T_G_C, const Pointcloud& points_C, const Colors& colors,
                       bool freespace_points) {
  CHECK_EQ(points_C.size(), colors.size());  // tsdf_integrator.cc:247
  // the reference's own tags (tsdf_integrator.cc:246, :311, :559), so that timing::Timing::Print() of an unmodified
  // voxblox_ros keeps its integrate/* rows; hip/* split the call into its three parts
  timing::Timer integrate_timer(kind == VBX_TSDF_SIMPLE ? "integrate/simple" : kind == VBX_TSDF_MERGED ? "integrate/merged" : "integrate/fast");
  MirrorRef pinned = mirrorOf(layer);
  DeviceMirror& dev = *pinned;
  timing::Timer reconcile_timer("hip/tsdf_reconcile_from_host");
  reconcileTsdfFromHost(dev, layer);  // removeDistantBlocks / loadMap / tsdfMapCallback since the last call
  reconcile_timer.Stop();
  const vbx_tsdf_cfg cfg = toC(config);
  const Point pos = T_G_C.getPosition();
  const auto& q = T_G_C.getRotation().toImplementation();  // Eigen::Quaternionf
  const float quat_wxyz[4] = {q.w(), q.x(), q.y(), q.z()};
  // AlignedVector<Eigen::Vector3f> is contiguous with a 12-byte stride; Color is 4 bytes
  static_assert(sizeof(Point) == 12 && sizeof(Color) == 4, "Pointcloud / Colors are handed over zero-copy");
  timing::Timer device_timer("hip/tsdf_integrate_device");
  CHECK_EQ(vbx_tsdf_integrate(dev.ctx, kind, &cfg, pos.data(), quat_wxyz, points_C.empty() ? nullptr : points_C[0].data(),
                              colors.empty() ? nullptr : &colors[0].r, points_C.size(), freespace_points ? 1 : 0),
           VBX_OK)
      << vbx_last_error(dev.ctx);
  if (kind == VBX_TSDF_FAST) ++dev.frames_integrated;
  device_timer.Stop();
  timing::Timer mirror_timer("hip/tsdf_mirror_to_host");
  mirrorTsdfToHost(dev, layer);
  mirror_timer.Stop();
  integrate_timer.Stop();
}
}  // namespace
}  // namespace hip

}  // namespace voxblox
extern "C" void vbx_dropin_set_reconcile_mode(int mode) { voxblox::hip::reconcileMode().store(mode); }
extern "C" int vbx_dropin_get_reconcile_mode() { return voxblox::hip::reconcileMode().load(); }
extern "C" void vbx_dropin_mark_tsdf_layer_edited(const void* tsdf_layer) {
  voxblox::hip::markLayerEdited(static_cast<const voxblox::Layer<voxblox::TsdfVoxel>*>(tsdf_layer));
}
extern "C" void vbx_dropin_mark_esdf_layer_edited(const void* esdf_layer) {
  voxblox::hip::markLayerEdited(static_cast<const voxblox::Layer<voxblox::EsdfVoxel>*>(esdf_layer));
}
namespace voxblox {

TsdfIntegratorBase::Ptr TsdfIntegratorFactory::create(const std::string& integrator_type_name,
                                                      const TsdfIntegratorBase::Config& config,
                                                      Layer<TsdfVoxel>* layer) {
  CHECK(!integrator_type_name.empty());
  int integrator_type = 1;
  for (const std::string& valid_integrator_type_name : kTsdfIntegratorTypeNames) {
    if (integrator_type_name == valid_integrator_type_name)
      return create(static_cast<TsdfIntegratorType>(integrator_type), config, layer);
    ++integrator_type;
  }
  LOG(FATAL) << "Unknown TSDF integrator type: " << integrator_type_name;
  return TsdfIntegratorBase::Ptr();
}

TsdfIntegratorBase::Ptr TsdfIntegratorFactory::create(const TsdfIntegratorType integrator_type,
                                                      const TsdfIntegratorBase::Config& config,
                                                      Layer<TsdfVoxel>* layer) {
  CHECK_NOTNULL(layer);
  switch (integrator_type) {
    case TsdfIntegratorType::kSimple: return TsdfIntegratorBase::Ptr(new SimpleTsdfIntegrator(config, layer));
    case TsdfIntegratorType::kMerged: return TsdfIntegratorBase::Ptr(new MergedTsdfIntegrator(config, layer));
    case TsdfIntegratorType::kFast: return TsdfIntegratorBase::Ptr(new FastTsdfIntegrator(config, layer));
    default: LOG(FATAL) << "Unknown TSDF integrator type: " << static_cast<int>(integrator_type); break;
  }
  return TsdfIntegratorBase::Ptr();
}

TsdfIntegratorBase::TsdfIntegratorBase(const Config& config, Layer<TsdfVoxel>* layer) : config_(config) {
  setLayer(layer);
  if (config_.integrator_threads == 0) {
    LOG(WARNING) << "Automatic core count failed, defaulting to 1 threads";
    config_.integrator_threads = 1;
  }
  // clearing rays have no utility if voxel_carving is disabled
  if (config_.allow_clear && !config_.voxel_carving_enabled) config_.allow_clear = false;
}

void TsdfIntegratorBase::setLayer(Layer<TsdfVoxel>* layer) {
  CHECK_NOTNULL(layer);
  layer_ = layer;
  voxel_size_ = layer_->voxel_size();
  block_size_ = layer_->block_size();
  voxels_per_side_ = layer_->voxels_per_side();
  voxel_size_inv_ = 1.0 / voxel_size_;
  block_size_inv_ = 1.0 / block_size_;
  voxels_per_side_inv_ = 1.0 / voxels_per_side_;
}

void SimpleTsdfIntegrator::integratePointCloud(const Transformation& T_G_C, const Pointcloud& points_C,
                                               const Colors& colors, const bool freespace_points) {
  hip::integrateOnDevice(VBX_TSDF_SIMPLE, config_, layer_, T_G_C, points_C, colors, freespace_points);
}

void MergedTsdfIntegrator::integratePointCloud(const Transformation& T_G_C, const Pointcloud& points_C,
                                               const Colors& colors, const bool freespace_points) {
  hip::integrateOnDevice(VBX_TSDF_MERGED, config_, layer_, T_G_C, points_C, colors, freespace_points);
}

void FastTsdfIntegrator::integratePointCloud(const Transformation& T_G_C, const Pointcloud& points_C,
                                             const Colors& colors, const bool freespace_points) {
  hip::integrateOnDevice(VBX_TSDF_FAST, config_, layer_, T_G_C, points_C, colors, freespace_points);
}

std::string TsdfIntegratorBase::Config::print() const {
  std::stringstream ss;
  ss << "================== TSDF Integrator Config (HIP drop-in) ====================\n"
     << " - default_truncation_distance: " << default_truncation_distance << "\n"
     << " - max_weight: " << max_weight << "\n"
     << " - voxel_carving_enabled: " << voxel_carving_enabled << "\n"
     << " - min_ray_length_m: " << min_ray_length_m << "\n"
     << " - max_ray_length_m: " << max_ray_length_m << "\n"
     << " - use_const_weight: " << use_const_weight << "\n"
     << " - allow_clear: " << allow_clear << "\n"
     << " - integration_order_mode: " << integration_order_mode << "\n"
     << "==============================================================================\n";
  return ss.str();
}

}  // namespace voxblox
