// esdf_order_model.cc — CPU model of the reference-order ESDF wavefront (design tool, not product code).
//
// Runs the oracle's Fast TSDF integration + the reference's incremental ESDF update frame by frame and
// instruments processOpenSet's pop sequence: pops per bucket, "generations" (the entries a bucket holds at the
// moment it becomes / stays the lowest non-empty one), pushes below the current bucket ("excursions": the
// reference pops those before it returns to the bucket), stale entries (value's bucket below the bucket the entry
// sits in) and entries with a usable neighbour of the other sign class.  Sizes the parallel replay before GPU time
// is spent on it.
//
//   g++ -O2 -std=c++17 -shared -fPIC -ffp-contract=off -I oracle tools/esdf_order_model.cc -o tools/libesdf_order_model.so
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <deque>
#include <map>
#include <memory>
#include <unordered_map>
#include <vector>

#include "vbx_esdf.hpp"
#include "vbx_tsdf.hpp"

// ---- the device replay (voxblox_amd/csrc/vbx_esdf_replay_core.hpp) run as serial loops over its thread ids:
// the HIP builtins it uses, for one host thread
#include <cmath>
#include <random>
template <typename T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> static inline T atomicAnd(T* p, T v) { T o = *p; *p = o & v; return o; }
template <typename T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <typename T> static inline T atomicCAS(T* p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }
static inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
#define RP_FN static inline
static int g_filter_level = 3;
#define RP_INC(p) atomicAdd((p), 1u)
#define RP_LD(x) (x)
#define RP_SHARD 0u
#define RP_LD64(x) (x)
#include "../voxblox_amd/csrc/vbx_esdf_replay_core.hpp"

using namespace orc;

namespace {

struct Stats {
  uint64_t pops = 0, skipped = 0, stale = 0, hazardous = 0, lower_pushes = 0, excursion_pops = 0, max_exc_depth = 0;
  uint64_t generations = 0, relax = 0, pushes = 0, same_bucket_pushes = 0;
  std::vector<uint64_t> gen_sizes;
  std::map<int, uint64_t> exc_size_hist, exc_pops_hist;  // excursion size (pops below the base bucket per root) -> count
  uint64_t raise_pops = 0;
  uint64_t base_by_bucket[32] = {0}, exc_by_bucket[32] = {0}, stale_by_bucket[32] = {0};
};

class ModelEsdf : public EsdfIntegrator {
 public:
  using EsdfIntegrator::EsdfIntegrator;
  Stats st;
  rp::Ctl emul_ctl;
  uint64_t emul_check_runs = 0, emul_check_failures = 0;   // EOM_CHECK: fixed points refolded in full / found inconsistent (last update)
  bool seq_watch = false;
  LIdx3 seq_watch_g{0, 0, 0};
  bool emul_shuffle = true;
  int mode = 0;  // 0: the reference's sequential wavefront (instrumented), 1: the batched event-fold replay
  size_t kmax = 1u << 20, smax = 256;
  int max_iters = 64;
  struct ParStats {
    uint64_t supersteps = 0, iterations = 0, folds = 0, records = 0, exc_records = 0, cut_iters = 0, cut_smax = 0, max_iters_seen = 0;
    uint64_t max_events = 0, small_steps = 0, small_iters = 0;  // supersteps with <= 1024 base records
    std::map<int, uint64_t> iters_hist;
  } ps;

  int bucketOf(double value) const {
    const double max_val = config_.max_distance_m;
    if (value > max_val) value = max_val;
    int b = static_cast<int>(std::floor(std::abs(value) / max_val * (config_.num_buckets - 1)));
    if (b >= config_.num_buckets) b = config_.num_buckets - 1;
    return b;
  }

  // the reference's update with the queue opened up for instrumentation
  void update(bool clear_updated_flag) {
    std::vector<Idx3> tsdf_blocks;
    tsdf_layer_->getAllUpdatedBlocks(kEsdf, &tsdf_blocks);
    tsdf_blocks.insert(tsdf_blocks.end(), updated_blocks_.begin(), updated_blocks_.end());   // esdf_integrator.cc:107-109
    updated_blocks_.clear();
    // classification through the base class with its own queues; we then drain them into ours.  The base class
    // calls processRaiseSet / processOpenSet itself, so replicate updateFromTsdfBlocks' tail here instead:
    // run the base with the real thing on a COPY?  Simpler: the base's methods are reused, the open set is replayed
    // by recording every push/pop through a shadow queue built from the same rules.
    base_update(tsdf_blocks);
    if (clear_updated_flag) {
      for (const Idx3& b : tsdf_blocks) {
        auto blk = tsdf_layer_->getBlockPtrByIndex(b);
        if (blk) blk->updated &= ~(1u << kEsdf);
      }
    }
  }

 private:
  std::vector<std::deque<LIdx3>> bq_;
  size_t n_open_ = 0;

  void qpush(const LIdx3& g, float v) {
    bq_[bucketOf(v)].push_back(g);
    ++n_open_;
    ++st.pushes;
  }

  void base_update(const std::vector<Idx3>& tsdf_blocks) {
    // updateFromTsdfBlocks' classification, copied in behaviour through the base class by temporarily capturing
    // its queue: the base class' open_ / raise_ are protected members, so drain them after classification.
    // To do that without running the base wavefront, call a trimmed copy of the loop.
    bq_.assign(config_.num_buckets, {});
    n_open_ = 0;
    // what addNewRobotPosition (the base class' own, esdf_integrator.cc:25-92) left in raise_ / open_ comes first.  open_
    // hands its entries out lowest bucket first, FIFO inside; an entry's bucket is that of its voxel's distance (pushed
    // with it at :84, and nothing has touched the voxel since — one position per update here)
    while (!raise_.empty()) { raise_q_.push_back(raise_.front()); raise_.pop(); }
    while (!open_.empty()) {
      const LIdx3 g = open_.front();
      open_.pop();
      qpush(g, esdf_layer_->getVoxelPtrByGlobalIndex(g)->distance);
    }
    classify(tsdf_blocks);
    if (std::getenv("EOM_WATCH")) {
      int bx, by, bz, lin;
      std::sscanf(std::getenv("EOM_WATCH"), "%d,%d,%d,%d", &bx, &by, &bz, &lin);
      seq_watch = true;
      seq_watch_g = globalVoxelIndexFromBlockAndVoxelIndex(Idx3{bx, by, bz}, Idx3{lin % 16, (lin / 16) % 16, lin / 256}, 16);
    }
    if (mode != 2) raiseSet();   // (the emulated device code pops raise_ itself)
    if (mode == 0) openSet(); else if (mode == 1) openSetParallel(); else openSetEmul();
  }

  void classify(const std::vector<Idx3>& tsdf_blocks) {
    const bool incremental = true;
    for (const Idx3& block_index : tsdf_blocks) {
      auto tsdf_block = tsdf_layer_->getBlockPtrByIndex(block_index);
      if (!tsdf_block) continue;
      auto esdf_block = esdf_layer_->allocateBlockPtrByIndex(block_index);
      esdf_block->updated = 0x1;
      const size_t num_voxels = tsdf_block->num_voxels;
      for (size_t lin = 0; lin < num_voxels; ++lin) {
        const TsdfVoxel& tsdf_voxel = tsdf_block->voxels[lin];
        if (tsdf_voxel.weight < config_.min_weight) continue;
        EsdfVoxel& esdf_voxel = esdf_block->voxels[lin];
        const Idx3 voxel_index = esdf_block->voxelIndexFromLinear(lin);
        const LIdx3 global_index =
            globalVoxelIndexFromBlockAndVoxelIndex(block_index, voxel_index, static_cast<int>(voxels_per_side_));
        const bool tsdf_fixed = isFixed(tsdf_voxel.distance);
        const float sd = signum(tsdf_voxel.distance) * config_.default_distance_m;
        if (!esdf_voxel.observed || esdf_voxel.hallucinated) {
          if (esdf_voxel.hallucinated) raise_q_.push_back(global_index);
          if (tsdf_fixed) {
            esdf_voxel.distance = tsdf_voxel.distance;
            esdf_voxel.fixed = true;
            esdf_voxel.in_queue = true;
            qpush(global_index, esdf_voxel.distance);
          } else {
            esdf_voxel.distance = sd;
            esdf_voxel.fixed = false;
            if (incremental && updateVoxelFromNeighbors(global_index)) {
              esdf_voxel.in_queue = true;
              qpush(global_index, esdf_voxel.distance);
            }
          }
          esdf_voxel.parent = {0, 0, 0};
        } else {
          if (tsdf_fixed || esdf_voxel.fixed) {
            if (!tsdf_fixed) {
              esdf_voxel.distance = sd;
              esdf_voxel.parent = {0, 0, 0};
              esdf_voxel.fixed = false;
              raise_q_.push_back(global_index);
              esdf_voxel.in_queue = true;
              qpush(global_index, esdf_voxel.distance);
            } else if ((esdf_voxel.distance > 0.0f && tsdf_voxel.distance + config_.min_diff_m < esdf_voxel.distance) ||
                       (esdf_voxel.distance <= 0.0f && tsdf_voxel.distance - config_.min_diff_m > esdf_voxel.distance)) {
              esdf_voxel.fixed = tsdf_fixed;
              esdf_voxel.distance = tsdf_voxel.distance;
              esdf_voxel.parent = {0, 0, 0};
              esdf_voxel.in_queue = true;
              qpush(global_index, esdf_voxel.distance);
            } else if ((esdf_voxel.distance > 0.0f && tsdf_voxel.distance - config_.min_diff_m > esdf_voxel.distance) ||
                       (esdf_voxel.distance <= 0.0f && tsdf_voxel.distance + config_.min_diff_m < esdf_voxel.distance)) {
              esdf_voxel.fixed = tsdf_fixed;
              esdf_voxel.distance = tsdf_voxel.distance;
              esdf_voxel.parent = {0, 0, 0};
              raise_q_.push_back(global_index);
              esdf_voxel.in_queue = true;
              qpush(global_index, esdf_voxel.distance);
            }
          } else if (signum(tsdf_voxel.distance) != signum(esdf_voxel.distance)) {
            if (tsdf_voxel.distance < esdf_voxel.distance) {
              esdf_voxel.distance = sd;
              esdf_voxel.parent = {0, 0, 0};
              esdf_voxel.in_queue = true;
              qpush(global_index, esdf_voxel.distance);
            } else {
              esdf_voxel.distance = sd;
              esdf_voxel.parent = {0, 0, 0};
              raise_q_.push_back(global_index);
            }
          }
        }
        esdf_voxel.observed = true;
        esdf_voxel.hallucinated = false;
      }
    }
  }

  std::deque<LIdx3> raise_q_;
  void raiseSet() {
    while (!raise_q_.empty()) {
      const LIdx3 g = raise_q_.front();
      raise_q_.pop_front();
      ++st.raise_pops;
      for (int idx = 0; idx < 26; ++idx) {
        const Idx3 off = NeighborhoodLut::offset(idx);
        const LIdx3 n{g.x + off.x, g.y + off.y, g.z + off.z};
        EsdfVoxel* nv = esdf_layer_->getVoxelPtrByGlobalIndex(n);
        if (nv == nullptr || !nv->observed || nv->fixed) continue;
        if (nv->parent == Idx3{-off.x, -off.y, -off.z}) {
          nv->distance = signum(nv->distance) * config_.default_distance_m;
          nv->parent = {0, 0, 0};
          raise_q_.push_back(n);
        } else if (!nv->in_queue) {
          qpush(n, nv->distance);
          nv->in_queue = true;
        }
      }
    }
  }


  // ---------------------------------------------------------------------------------------------------------
  // The batched replay.  A super-step takes the FIFO prefix of the lowest non-empty bucket b as BASE records; a
  // record is one pop.  Pops whose pushes land below b spawn EXCURSION records (the reference pops those before
  // it returns to b).  Every voxel next to a record folds the events that concern it — offers from records on
  // its 26 neighbours, its own pops — in pop-time order T, with each record's voxel state at its pop time taken
  // from the previous iteration; iterate to the fixed point (the causal order makes the stable prefix exact),
  // commit the stable prefix.
  struct Rec {
    EsdfVoxel* v;
    LIdx3 g;
    int pusher, lut, bucket;
    bool live;
    uint64_t T;
    float sd;      // voxel distance at pop time (speculated)
    Idx3 sp;       // parent at pop time
  };
  struct Push { uint64_t T; int lut; int bucket; LIdx3 g; int rec; };
  static int oppositeLut(int idx) {
    static int opp[26];
    static bool init = false;
    if (!init) {
      for (int i = 0; i < 26; ++i)
        for (int j = 0; j < 26; ++j) {
          const Idx3 a = NeighborhoodLut::offset(i), c = NeighborhoodLut::offset(j);
          if (a.x == -c.x && a.y == -c.y && a.z == -c.z) opp[i] = j;
        }
      init = true;
    }
    return opp[idx];
  }
  struct Target {
    EsdfVoxel* v;
    LIdx3 g;
    EsdfVoxel* nb[26];
  };

  // the relaxation rule of processOpenSet for one (popped voxel state, neighbour state, lut) — esdf_integrator.cc:405-491
  bool relaxRule(float vd, float nd, int lut, float* out) const {
    const float distance = NeighborhoodLut::distance(lut) * voxel_size_;
    if (vd > 0 && nd > 0) {
      if (vd + distance + config_.min_diff_m < nd) { *out = vd + distance; return true; }
    } else if (vd <= 0 && nd <= 0) {
      if (vd - distance - config_.min_diff_m > nd) { *out = vd - distance; return true; }
    } else {
      const float potential = vd - signum(vd) * distance;
      if (std::abs(potential - nd) > distance) {
        if (static_cast<float>(signum(potential)) == nd) *out = potential;
        else *out = signum(nd) * distance;
        return true;
      }
    }
    return false;
  }

  struct Tgt {
    EsdfVoxel* v;
    LIdx3 g;
    int nb[26];              // target index of the neighbour voxel, -1: not a target (no records there), -2: no such voxel
    std::vector<int> recs;   // records on this voxel
    bool dirty;
  };

  void openSetParallel() {
    const bool trace = std::getenv("EOM_TRACE") != nullptr;
    while (n_open_ != 0) {
      int b = 0;
      while (bq_[b].empty()) ++b;
      const size_t K = std::min(bq_[b].size(), kmax);
      std::vector<Rec> recs;
      recs.reserve(K * 2);
      std::unordered_map<EsdfVoxel*, int> tgt_of;
      std::vector<Tgt> tg;
      std::vector<int> dirty_list;
      auto markDirty = [&](int t) { if (!tg[t].dirty) { tg[t].dirty = true; dirty_list.push_back(t); } };
      auto addTarget = [&](EsdfVoxel* v, const LIdx3& g) -> int {
        auto it = tgt_of.find(v);
        if (it != tgt_of.end()) return it->second;
        const int id = (int)tg.size();
        tg.emplace_back();
        Tgt& t = tg.back();
        t.v = v; t.g = g; t.dirty = false;
        tgt_of[v] = id;
        for (int i = 0; i < 26; ++i) {
          const Idx3 o = NeighborhoodLut::offset(i);
          EsdfVoxel* nv = esdf_layer_->getVoxelPtrByGlobalIndex({g.x + o.x, g.y + o.y, g.z + o.z});
          if (nv == nullptr) { t.nb[i] = -2; continue; }
          auto jt = tgt_of.find(nv);
          if (jt == tgt_of.end()) { t.nb[i] = -1; continue; }
          t.nb[i] = jt->second;
          tg[jt->second].nb[oppositeLut(i)] = id;
        }
        return id;
      };
      // a record's voxel and its 26 neighbours are targets; all of them must be (re)folded
      auto placeRecord = [&](int r) {
        const LIdx3 g = recs[r].g;
        const int t = addTarget(recs[r].v, g);
        tg[t].recs.push_back(r);
        markDirty(t);
        for (int i = 0; i < 26; ++i) {
          if (tg[t].nb[i] == -2) continue;
          const Idx3 o = NeighborhoodLut::offset(i);
          const LIdx3 ng{g.x + o.x, g.y + o.y, g.z + o.z};
          int u = tg[t].nb[i];
          if (u == -1) u = addTarget(esdf_layer_->getVoxelPtrByGlobalIndex(ng), ng);
          markDirty(u);
        }
      };
      auto dirtyAround = [&](int r) {
        const int t = tgt_of.at(recs[r].v);
        markDirty(t);
        for (int i = 0; i < 26; ++i)
          if (tg[t].nb[i] >= 0) markDirty(tg[t].nb[i]);
      };
      for (size_t i = 0; i < K; ++i) {
        const LIdx3 g = bq_[b][i];
        EsdfVoxel* v = esdf_layer_->getVoxelPtrByGlobalIndex(g);
        recs.push_back({v, g, -1, 0, b, true, (uint64_t)i << 24, v->distance, v->parent});
      }
      for (size_t i = 0; i < K; ++i) placeRecord((int)i);
      std::unordered_map<uint64_t, int> child;  // (pusher << 5 | lut) -> record

      struct Ev { uint64_t T; int rec; int lut; };  // lut < 0: own pop
      struct SpecOut { int rec; float d; Idx3 p; };
      auto fold = [&](const Tgt& t, uint64_t limit, std::vector<SpecOut>* spec_out, std::vector<Push>* pushes, float* d_out,
                      Idx3* p_out, bool* q_out) {
        static thread_local std::vector<Ev> evv;
        evv.resize(4096);
        Ev* ev = evv.data();
        const int kCap = 4096;
        int n = 0;
        for (int i = 0; i < 26; ++i) {
          if (t.nb[i] < 0) continue;
          for (int r : tg[t.nb[i]].recs)
            if (recs[r].live && recs[r].T < limit && n < kCap) ev[n++] = {recs[r].T, r, oppositeLut(i)};
        }
        for (int r : t.recs)
          if (recs[r].live && recs[r].T < limit && n < kCap) ev[n++] = {recs[r].T, r, -1};
        if (n >= kCap) { std::fprintf(stderr, "event overflow\n"); std::abort(); }
        ps.max_events = std::max<uint64_t>(ps.max_events, n);
        std::sort(ev, ev + n, [](const Ev& x, const Ev& y) { return x.T < y.T; });
        float d = t.v->distance;
        Idx3 par = t.v->parent;
        bool inq = t.v->in_queue;
        for (int k = 0; k < n; ++k) {
          const Ev& e = ev[k];
          const Rec& r = recs[e.rec];
          if (e.lut < 0) {
            if (spec_out) spec_out->push_back({e.rec, d, par});
            inq = false;
            continue;
          }
          if (!r.v->observed || r.sd >= config_.max_distance_m || r.sd <= -config_.max_distance_m) continue;
          if (!t.v->observed || t.v->fixed) continue;
          float nd;
          if (relaxRule(r.sd, d, e.lut, &nd)) {
            const Idx3 o = NeighborhoodLut::offset(e.lut);
            d = nd;
            par = {-o.x, -o.y, -o.z};
            if (config_.multi_queue || !inq) {
              inq = true;
              if (pushes) pushes->push_back({r.T, e.lut, bucketOf(nd), t.g, e.rec});
            }
          }
        }
        *d_out = d; *p_out = par; *q_out = inq;
      };

      uint64_t cut = ~0ull;  // records with T < cut are final
      int it_count = 0;
      uint64_t folds_here = 0;
      for (;;) {
        ++it_count;
        ++ps.iterations;
        std::vector<int> todo;
        todo.swap(dirty_list);
        for (int t : todo) tg[t].dirty = false;
        // Jacobi: every fold of this iteration reads the records as the previous iteration left them
        struct LiveOut { int rec; bool live; int bucket; };
        std::vector<SpecOut> spec;
        std::vector<LiveOut> liveness;
        std::vector<Push> born;
        std::vector<Push> pushes;
        for (int ti : todo) {
          const Tgt& t = tg[ti];
          float d; Idx3 p; bool q;
          pushes.clear();
          fold(t, ~0ull, &spec, &pushes, &d, &p, &q);
          ++ps.folds;
          ++folds_here;
          // excursion records on this voxel: alive iff this fold pushed them below b
          for (int r : t.recs) {
            if (recs[r].pusher < 0) continue;
            bool found = false;
            int bucket = recs[r].bucket;
            for (const Push& pu : pushes)
              if (pu.bucket < b && pu.rec == recs[r].pusher && pu.lut == recs[r].lut) { found = true; bucket = pu.bucket; }
            if (found != recs[r].live || bucket != recs[r].bucket) liveness.push_back({r, found, bucket});
          }
          for (const Push& pu : pushes)
            if (pu.bucket < b && !child.count(((uint64_t)pu.rec << 5) | (uint64_t)pu.lut)) born.push_back(pu);
        }
        std::vector<int> change_recs;    // records whose pop-time state / pushes changed (the time of the change is their T)
        std::vector<int> resim;          // base records whose subtree changed
        for (const SpecOut& so : spec) {
          Rec& r = recs[so.rec];
          if (r.sd != so.d || !(r.sp == so.p)) {
            r.sd = so.d; r.sp = so.p;
            change_recs.push_back(so.rec);
            dirtyAround(so.rec);
          }
        }
        for (const LiveOut& lo : liveness) {
          recs[lo.rec].live = lo.live;
          recs[lo.rec].bucket = lo.bucket;
          change_recs.push_back(recs[lo.rec].pusher);
          dirtyAround(lo.rec);
          resim.push_back(lo.rec);
        }
        for (const Push& pu : born) {
          EsdfVoxel* v = esdf_layer_->getVoxelPtrByGlobalIndex(pu.g);
          const int id = (int)recs.size();
          recs.push_back({v, pu.g, pu.rec, pu.lut, pu.bucket, true, ~0ull, v->distance, v->parent});
          child[((uint64_t)pu.rec << 5) | (uint64_t)pu.lut] = id;
          placeRecord(id);
          change_recs.push_back(pu.rec);
          resim.push_back(id);
          ++ps.exc_records;
        }
        // pop times of the subtrees that changed: the reference's queue discipline over the live excursion records
        uint64_t smax_cut = ~0ull;
        if (!resim.empty()) {
          std::vector<char> base_dirty(K, 0);
          for (int r : resim) {
            int a = r;
            while (recs[a].pusher >= 0) a = recs[a].pusher;
            base_dirty[a] = 1;
          }
          std::vector<std::vector<int>> kids(recs.size());
          for (size_t r = K; r < recs.size(); ++r) {
            if (!recs[r].live) continue;
            kids[recs[r].pusher].push_back((int)r);
          }
          for (size_t r = K; r < recs.size(); ++r) {
            int a = (int)r;
            while (recs[a].pusher >= 0) a = recs[a].pusher;
            if (base_dirty[a]) recs[r].T = ~0ull;
          }
          for (auto& k : kids) std::sort(k.begin(), k.end(), [&](int x, int y) { return recs[x].lut < recs[y].lut; });
          for (size_t i = 0; i < K; ++i) {
            if (!base_dirty[i] || kids[i].empty()) continue;
            std::vector<std::deque<int>> q(b);
            uint64_t rank = 0;
            auto pushKids = [&](int r) { for (int c : kids[r]) q[recs[c].bucket].push_back(c); };
            pushKids((int)i);
            for (;;) {
              int lb = 0;
              while (lb < b && q[lb].empty()) ++lb;
              if (lb == b) break;
              const int r = q[lb].front();
              q[lb].pop_front();
              recs[r].T = ((uint64_t)i << 24) | ++rank;
              if (rank >= smax) break;   // the rest of this excursion waits for a later super-step
              pushKids(r);
            }
          }
        }
        for (size_t r = K; r < recs.size(); ++r)
          if (recs[r].live && recs[r].T != ~0ull && (recs[r].T & 0xFFFFFF) == smax) smax_cut = std::min(smax_cut, recs[r].T);
        uint64_t first_change = ~0ull;
        for (int r : change_recs) first_change = std::min(first_change, recs[r].T);
        if (first_change >= smax_cut || it_count >= max_iters) {  // everything in front of the bound is a fixed point
          cut = std::min(first_change, smax_cut);
          if (cut != ~0ull) { if (smax_cut <= first_change) ++ps.cut_smax; else ++ps.cut_iters; }
          break;
        }
      }
      if (trace) std::fprintf(stderr, "superstep b=%d K=%zu recs=%zu targets=%zu iters=%d folds=%llu cut=%llx\n", b, K, recs.size(), tg.size(), it_count, (unsigned long long)folds_here, (unsigned long long)cut);
      ps.iters_hist[it_count]++;
      ps.max_iters_seen = std::max<uint64_t>(ps.max_iters_seen, it_count);
      ++ps.supersteps;
      if (K <= 1024) { ++ps.small_steps; ps.small_iters += it_count; }

      // ---- commit everything in front of `cut`
      std::vector<Push> pushes;
      struct Final { EsdfVoxel* v; float d; Idx3 p; bool q; };
      std::vector<Final> finals;
      for (const Tgt& t : tg) {
        float d; Idx3 p; bool q;
        fold(t, cut, nullptr, &pushes, &d, &p, &q);
        finals.push_back({t.v, d, p, q});
      }
      for (const Final& f : finals) { f.v->distance = f.d; f.v->parent = f.p; f.v->in_queue = f.q; }
      std::sort(pushes.begin(), pushes.end(), [](const Push& x, const Push& y) { return x.T != y.T ? x.T < y.T : x.lut < y.lut; });
      size_t committed_base = 0;
      for (size_t i = 0; i < K; ++i)
        if (recs[i].T < cut) ++committed_base;
      for (size_t r = 0; r < recs.size(); ++r)
        if (recs[r].live && recs[r].T < cut) { ++st.pops; ++ps.records; }
      for (size_t i = 0; i < committed_base; ++i) bq_[b].pop_front();
      n_open_ -= committed_base;
      for (const Push& pu : pushes) {
        if (pu.bucket < b) {
          const int c = child.at(((uint64_t)pu.rec << 5) | (uint64_t)pu.lut);
          if (recs[c].live && recs[c].T < cut) continue;  // popped inside this super-step
        }
        bq_[pu.bucket].push_back(pu.g);
        ++n_open_;
      }
    }
  }

  // ---------------------------------------------------------------------------------------------------------
  // mode 2: the device code itself (rp:: phases + control), every phase as a loop over its thread ids in a
  // shuffled order, on flat copies of the layer and of the queue
  void openSetEmul() {
    using namespace rp;
    // blocks -> pool slots
    std::vector<Idx3> blocks;
    for (auto& kv : esdf_layer_->block_map) blocks.push_back(kv.first);
    std::unordered_map<Idx3, uint32_t, AnyIndexHasher> slot_of;
    for (size_t i = 0; i < blocks.size(); ++i) slot_of[blocks[i]] = (uint32_t)i;
    const uint32_t nvox = 4096, vps = 16;
    const size_t nv = blocks.size() * nvox;
    std::vector<float> dist(nv);
    std::vector<uint32_t> state(nv), nbslot(blocks.size() * 27, kNone);
    for (size_t sl = 0; sl < blocks.size(); ++sl) {
      auto& blk = *esdf_layer_->block_map[blocks[sl]];
      for (uint32_t i = 0; i < nvox; ++i) {
        const EsdfVoxel& v = blk.voxels[i];
        dist[sl * nvox + i] = v.distance;
        state[sl * nvox + i] = (v.observed ? kObserved : 0) | (v.hallucinated ? kHallucinated : 0) | (v.in_queue ? kInQueue : 0) |
                               (v.fixed ? kFixed : 0) | rp_pack_parent(v.parent.x, v.parent.y, v.parent.z);
      }
      for (int k = 0; k < 27; ++k) {
        const Idx3 nb{blocks[sl].x + (k % 3) - 1, blocks[sl].y + (k / 3 % 3) - 1, blocks[sl].z + (k / 9) - 1};
        auto it = slot_of.find(nb);
        if (it != slot_of.end()) nbslot[sl * 27 + k] = it->second;
      }
    }
    auto gidOf = [&](const LIdx3& g) {
      const Idx3 b = blockIndexFromGlobalVoxelIndex(g, 1.0f / vps);
      const Idx3 l = localFromGlobalVoxelIndex(g, vps);
      return slot_of.at(b) * nvox + (uint32_t)(l.x + vps * (l.y + vps * l.z));
    };
    // Args::hazard: a neighbour of the other sign class exists
    std::vector<uint8_t> hazard(nv, 0);
    if (!std::getenv("EOM_NO_FILTER")) {
      for (size_t sl = 0; sl < blocks.size(); ++sl)
        for (uint32_t i = 0; i < nvox; ++i) {
          const LIdx3 g = globalVoxelIndexFromBlockAndVoxelIndex(blocks[sl], esdf_layer_->block_map[blocks[sl]]->voxelIndexFromLinear(i), vps);
          const bool pos = dist[sl * nvox + i] > 0;
          for (int k = 0; k < 26; ++k) {
            const Idx3 o = NeighborhoodLut::offset(k);
            EsdfVoxel* nvx = esdf_layer_->getVoxelPtrByGlobalIndex({g.x + o.x, g.y + o.y, g.z + o.z});
            if (nvx && nvx->observed && ((nvx->distance > 0) != pos)) { hazard[sl * nvox + i] = 1; break; }
          }
        }
    }
    g_filter_level = std::getenv("EOM_FILTER_LEVEL") ? std::atoi(std::getenv("EOM_FILTER_LEVEL")) : 3;
    Args a{};
    a.hazard = std::getenv("EOM_NO_FILTER") ? nullptr : hazard.data();
    a.c.filter = (uint32_t)g_filter_level;
    a.c.stats = 1;   // (steps per phase: printed below)
    a.c.mark_moved = std::getenv("EOM_NO_MARK_MOVED") ? 0u : (std::getenv("EOM_MARK_MOVED") ? (uint32_t)std::atoi(std::getenv("EOM_MARK_MOVED")) : 2u);
    a.c.fold_all = std::getenv("EOM_NO_FOLD_ALL") ? 0u : 1u;
    a.c.ev = std::getenv("EOM_EV") ? (uint32_t)std::atoi(std::getenv("EOM_EV")) : 256u;
    if (a.c.ev > kEvMax) a.c.ev = kEvMax;
    a.c.tgt_claim = std::getenv("EOM_NO_TGT_CLAIM") ? 0u : 1u;
    a.c.slot_by_base = std::getenv("EOM_NO_SLOT_BY_BASE") ? 0u : 1u;
    a.c.member_limit = std::getenv("EOM_MEMBER_LIMIT") ? 1u : 0u;
    a.c.max_distance = config_.max_distance_m; a.c.min_diff = config_.min_diff_m; a.c.voxel_size = voxel_size_; a.c.default_distance = config_.default_distance_m;
    a.c.full = config_.full_euclidean_distance; a.c.multi_queue = config_.multi_queue; a.c.num_buckets = config_.num_buckets;
    a.c.kmax = (uint32_t)std::min<size_t>(kmax, 1u << 20); a.c.smax = (uint32_t)smax; a.c.max_iters = (uint32_t)max_iters;
    a.c.cut_mult = 2; a.c.ramp_mult = 4;
    Ctl& c = emul_ctl;
    c = Ctl{};
    a.ctl = &c;
    a.dist = dist.data(); a.state = state.data(); a.nbslot = nbslot.data(); a.blk_dirty = nullptr; a.nvox = nvox; a.vps = vps;
    const uint32_t max_chunks = (uint32_t)(8 * nv / kChunk + 64);
    std::vector<uint32_t> arena((size_t)max_chunks * kChunk), chunk_tab((size_t)(config_.num_buckets + 1) * max_chunks);
    a.arena = arena.data(); a.chunk_tab = chunk_tab.data(); a.max_chunks = max_chunks;
    for (int b = 0; b < config_.num_buckets; ++b) {
      rp_queue_reserve(a, b, (uint32_t)bq_[b].size());
      for (size_t i = 0; i < bq_[b].size(); ++i) rp_queue_store(a, b, (uint32_t)i, gidOf(bq_[b][i]));
      c.tail[b] = (uint32_t)bq_[b].size();
      bq_[b].clear();
    }
    {
      const int RQ = config_.num_buckets;
      rp_queue_reserve(a, RQ, (uint32_t)raise_q_.size());
      for (size_t i = 0; i < raise_q_.size(); ++i) rp_queue_store(a, RQ, (uint32_t)i, gidOf(raise_q_[i]));
      c.tail[RQ] = (uint32_t)raise_q_.size();
      raise_q_.clear();
    }
    n_open_ = 0;
    const uint32_t rec_cap = std::getenv("EOM_REC_CAP") ? (uint32_t)std::atoi(std::getenv("EOM_REC_CAP")) : a.c.kmax * 8 + 65536;
    const uint32_t tgt_cap = std::getenv("EOM_TGT_CAP") ? (uint32_t)std::atoi(std::getenv("EOM_TGT_CAP")) : rec_cap * 4;
    a.rec_cap = rec_cap; a.tgt_cap = tgt_cap;
    std::vector<uint32_t> rec_vox(rec_cap), rec_pusher(rec_cap), rec_base(rec_cap), rec_meta(rec_cap), rec_meta_n(rec_cap), rec_poison(rec_cap),
        rec_s(rec_cap), rec_s_n(rec_cap), rec_kid((size_t)rec_cap * 26), rec_tgts((size_t)rec_cap * 27), rec_push((size_t)rec_cap * 7);
    std::vector<unsigned long long> rec_T(rec_cap);
    std::vector<float> rec_d(rec_cap), rec_d_n(rec_cap);
    a.rec_vox = rec_vox.data(); a.rec_pusher = rec_pusher.data(); a.rec_base = rec_base.data(); a.rec_meta = rec_meta.data();
    a.rec_meta_n = rec_meta_n.data(); a.rec_poison = rec_poison.data(); a.rec_T = rec_T.data(); a.rec_d = rec_d.data(); a.rec_d_n = rec_d_n.data();
    a.rec_s = rec_s.data(); a.rec_s_n = rec_s_n.data(); a.rec_kid = rec_kid.data(); a.rec_tgts = rec_tgts.data(); a.rec_push = rec_push.data();
    std::vector<uint32_t> vox2tgt(nv), tgt_gid(tgt_cap), tgt_cnt(tgt_cap), tgt_ev((size_t)tgt_cap * kEvMax), tgt_dirty(tgt_cap), dl0(tgt_cap), dl1(tgt_cap);
    a.vox2tgt = vox2tgt.data(); a.tgt_gid = tgt_gid.data(); a.tgt_cnt = tgt_cnt.data(); a.tgt_ev = tgt_ev.data(); a.tgt_dirty = tgt_dirty.data();
    a.dl[0] = dl0.data(); a.dl[1] = dl1.data();
    std::vector<uint32_t> chg(rec_cap), born((size_t)rec_cap * 6), cp((size_t)rec_cap * 4), sd_list(a.c.kmax), sub_dirty(a.c.kmax), sub_n(a.c.kmax), sub_slot(a.c.kmax);
    const uint32_t sub_slots_cap = std::max<uint32_t>(a.c.kmax, 4096);
    std::vector<uint32_t> sub_list((size_t)sub_slots_cap * a.c.smax), ord(rec_cap), off0(a.c.kmax);
    std::vector<unsigned long long> sim_q((size_t)sub_slots_cap * a.c.smax);
    uint32_t sub_slots_used = 0;
    a.chg = chg.data(); a.born = born.data(); a.cp = cp.data(); a.sd_list = sd_list.data(); a.sub_dirty = sub_dirty.data(); a.sub_n = sub_n.data();
    a.sub_slot = sub_slot.data(); a.sub_list = sub_list.data(); a.sub_slots_used = &sub_slots_used; a.sim_q = sim_q.data(); a.sub_slots_cap = sub_slots_cap;
    a.ord = ord.data(); a.off0 = off0.data();
    std::vector<uint32_t> sim_old((size_t)sub_slots_cap * a.c.smax);
    a.sim_old = sim_old.data();
    // the member lists PH_APPLY keeps for the device's ranking (the serial ranking does not read them: kept here so that the
    // code that fills them runs, and is bounds-checked, without a GPU)
    std::vector<uint32_t> sub_mem((size_t)sub_slots_cap * a.c.smax), sub_mem_n(a.c.kmax), rec_local(rec_cap), sub_restart(a.c.kmax, kNone);
    a.sub_mem = sub_mem.data(); a.sub_mem_n = sub_mem_n.data(); a.rec_local = rec_local.data(); a.sub_restart = sub_restart.data();
    std::vector<uint32_t> rec_born_it(rec_cap);
    a.rec_born_it = std::getenv("EOM_NO_BORN_IT") ? nullptr : rec_born_it.data();

    uint32_t watch = kNone;
    if (std::getenv("EOM_WATCH")) { int bx, by, bz, lin; std::sscanf(std::getenv("EOM_WATCH"), "%d,%d,%d,%d", &bx, &by, &bz, &lin); watch = slot_of.at(Idx3{bx, by, bz}) * nvox + (uint32_t)lin; }
    uint64_t check_failures = 0, check_runs = 0;
    std::mt19937 rng(12345);
    std::vector<uint32_t> order;
    c.phase = PH_BEGIN;
    rp_control(a);
    uint64_t steps = 0;
    while (!c.done) {
      ++steps;
      const uint32_t n = c.n_threads;
      if (c.phase == PH_RANK || c.phase == PH_PUSH) {
        Cnt4 run{};
        const uint32_t n_scan = c.phase == PH_PUSH ? c.scan_n : n;
        for (uint32_t i = 0; i < n_scan; ++i) {
          const Cnt4 cnt = rp_scan_count(a, i);
          rp_scan_apply(a, i, run);
          for (int k = 0; k < kScanC; ++k) run.v[k] += cnt.v[k];
        }
        for (int k = 0; k < kScanC; ++k) c.scan_tot[k] = run.v[k];
        if (c.phase == PH_PUSH && c.push_last)   // the last pass cleans up behind itself
          for (uint32_t tid = 0; tid < n; ++tid) rp_phase_cleanup(a, tid);
      } else {
        order.resize(n);
        for (uint32_t i = 0; i < n; ++i) order[i] = i;
        if (emul_shuffle) std::shuffle(order.begin(), order.end(), rng);
        for (uint32_t k = 0; k < n; ++k) {
          const uint32_t tid = order[k];
          switch (c.phase) {
            case PH_PLACE_BASE: rp_phase_place_base(a, tid); break;
            case PH_FOLD: rp_phase_fold(a, tid); break;
            case PH_APPLY: rp_phase_apply(a, tid); break;
            case PH_SIM: rp_phase_sim(a, tid); break;
            case PH_MINCUT: rp_phase_mincut(a, tid); break;
            case PH_COMMIT_FOLD: rp_phase_commit_fold(a, tid); break;
            case PH_RANK_WRITE: rp_phase_rank_write(a, tid); break;
            case PH_CLEANUP: rp_phase_cleanup(a, tid); break;
            case PH_RAISE_FOLD: rp_phase_raise_fold(a, tid); break;
            default: std::fprintf(stderr, "bad phase %u\n", c.phase); std::abort();
          }
        }
      }
      if (watch != kNone) {
        static uint32_t last_s = 0xdeadbeef; static float last_d = -1e9f;
        if (state[watch] != last_s || dist[watch] != last_d) {
          std::fprintf(stderr, "[watch] after phase %u (superstep %llu b=%u K=%u cut=%llx iter=%u): d=%.9g s=%08x\n", c.phase, c.st_supersteps, c.bucket, c.K, c.cut, c.iter, dist[watch], state[watch]);
          last_s = state[watch]; last_d = dist[watch];
        }
        if (c.phase == PH_COMMIT_FOLD && a.vox2tgt[watch]) {
          const uint32_t t = a.vox2tgt[watch] - 1;
          std::fprintf(stderr, "[watch] commit of superstep %llu: target %u events %u\n", c.st_supersteps, t, a.tgt_cnt[t]);
          for (uint32_t k = 0; k < a.tgt_cnt[t] && k < a.c.ev; ++k) {
            const uint32_t code = a.tgt_ev[(size_t)t * a.c.ev + k], r = code >> 5, lut = code & 31;
            uint32_t pv = lut < 26 ? ((a.rec_push[r * 7 + lut / 4] >> ((lut % 4) * 8)) & 0xFF) : 0;
            const uint32_t rg = a.rec_vox[r];
            const Idx3 rb = blocks[rg / nvox];
            const uint32_t rl = rg % nvox;
            std::fprintf(stderr, "[watch]   ev rec %u vox (%d %d %d) lut %u T=%llx live=%d poison=%u d=%.9g s=%08x push=%u kid=%u\n", r, rb.x * 16 + (int)(rl % 16), rb.y * 16 + (int)(rl / 16 % 16), rb.z * 16 + (int)(rl / 256), lut, a.rec_T[r], (int)rp_meta_live(a.rec_meta[r]), a.rec_poison[r],
                         a.rec_d[r], a.rec_s[r], pv, lut < 26 ? a.rec_kid[(size_t)r * 26 + lut] : 0);
          }
        }
        if (c.phase == PH_PLACE_BASE)
          for (uint32_t r = 0; r < c.K; ++r) if (a.rec_vox[r] == watch) std::fprintf(stderr, "[watch] base record %u of superstep %llu b=%u\n", r, c.st_supersteps, c.bucket);
        if (c.phase == PH_CLEANUP) {
          for (uint32_t r = 0; r < c.n_rec; ++r) if (a.rec_vox[r] == watch) std::fprintf(stderr, "[watch] record %u (pusher %u) T=%llx live=%d poison=%u cut=%llx\n", r, a.rec_pusher[r], a.rec_T[r], (int)rp_meta_live(a.rec_meta[r]), a.rec_poison[r], c.cut);
          for (int q = 0; q <= config_.num_buckets; ++q)
            for (uint32_t i = c.head[q]; i < c.tail[q]; ++i) if (rp_queue_entry(a, q, i) == watch) std::fprintf(stderr, "[watch] queued in %d at %u (head %u tail %u)\n", q, i, c.head[q], c.tail[q]);
        }
      }
      if (c.phase == PH_CLEANUP && std::getenv("EOM_SLOTS")) {   // excursions (base records with a ranking list) of the super-step
        uint32_t n_exc = 0;
        for (uint32_t i = 0; i < c.K; ++i) n_exc += a.sub_slot[i] != 0u;
        if (n_exc > 1000) std::fprintf(stderr, "[slots] superstep %llu b=%u K=%u: %u excursions, %u records\n", c.st_supersteps, c.bucket, c.K, n_exc, c.n_rec);
      }
      if (std::getenv("EOM_TRACE2")) std::fprintf(stderr, "phase %u n=%u iter=%u rec=%u tgt=%u\n", c.phase, n, c.iter, c.n_rec, c.n_tgt);
      const bool was_fold = c.phase == PH_FOLD && c.chg_n[0].v == 0 && c.born_n[0].v == 0;

      if (std::getenv("EOM_TRACE") && c.phase == PH_CLEANUP) std::fprintf(stderr, "superstep b=%u K=%u recs=%u tgts=%u iters=%u cut=%llx commit=%u\n", c.bucket, c.K, c.n_rec, c.a_tgt, c.iter, c.cut, c.n_commit);
      rp_control(a);
      // EOM_CHECK: at a fixed point (a FOLD phase that changed nothing, the cut is known now) a fold of EVERY target must change
      // nothing in front of the cut either — a target whose events moved relative to each other without being marked dirty
      // would show up here
      if (was_fold && c.phase == PH_COMMIT_FOLD && std::getenv("EOM_CHECK")) {
        const uint32_t nt = c.n_tgt < a.tgt_cap ? c.n_tgt : a.tgt_cap;
        for (uint32_t t = 0; t < nt; ++t) rp_fold(a, t, kNever, false);
        uint32_t bad_chg = 0, bad_born = 0;
        for (uint32_t k = 0; k < c.chg_n[0].v; ++k) {
          const uint32_t r = a.chg[k];
          unsigned long long T = a.rec_T[r];
          if (a.rec_pusher[r] != kNone && (a.rec_meta_n[r] & ~(1u << 18)) != (a.rec_meta[r] & ~(1u << 18))) T = a.rec_T[a.rec_pusher[r]];   // liveness: decided at the pusher's pop
          if (T < c.cut) ++bad_chg;
        }
        for (uint32_t k = 0; k < c.born_n[0].v && k < a.rec_cap; ++k) if (a.rec_T[a.born[(size_t)k * 6]] < c.cut) ++bad_born;
        if (bad_chg || bad_born)
          std::fprintf(stderr, "[check] superstep %llu b=%u K=%u iter=%u cut=%llx: a full refold at the fixed point changes %u records, bears %u in front of the cut (%u / %u anywhere)\n", c.st_supersteps, c.bucket, c.K, c.iter, c.cut, bad_chg, bad_born, c.chg_n[0].v, c.born_n[0].v);
        if (bad_chg || bad_born) ++check_failures;
        ++check_runs;
        c.chg_n[0].v = c.born_n[0].v = 0;
      }
    }
    if (c.error) std::fprintf(stderr, "EMUL ERROR %u\n", c.error);
    if (check_runs) std::fprintf(stderr, "[check] %llu fixed points refolded in full, %llu inconsistent\n", (unsigned long long)check_runs, (unsigned long long)check_failures);
    emul_check_runs = check_runs;
    emul_check_failures = check_failures;
    st.pops += c.st_pops;
    // back into the layer
    for (size_t sl = 0; sl < blocks.size(); ++sl) {
      auto& blk = *esdf_layer_->block_map[blocks[sl]];
      for (uint32_t i = 0; i < nvox; ++i) {
        EsdfVoxel& v = blk.voxels[i];
        const uint32_t s = state[sl * nvox + i];
        v.distance = dist[sl * nvox + i];
        v.in_queue = (s & kInQueue) != 0;
        int px, py, pz;
        rp_unpack_parent(s, &px, &py, &pz);
        v.parent = {px, py, pz};
      }
    }
  }

  bool hazardous(const LIdx3& g, const EsdfVoxel& v) {
    for (int idx = 0; idx < 26; ++idx) {
      const Idx3 off = NeighborhoodLut::offset(idx);
      EsdfVoxel* nv = esdf_layer_->getVoxelPtrByGlobalIndex({g.x + off.x, g.y + off.y, g.z + off.z});
      if (nv == nullptr || !nv->observed || nv->fixed) continue;
      if ((v.distance > 0) != (nv->distance > 0)) return true;
    }
    return false;
  }

  void openSet() {
    int base_bucket = -1;       // the bucket whose generation is being worked on
    uint64_t gen_left = 0;      // entries of the current generation not yet popped
    uint64_t exc_pops_this_root = 0;
    bool in_exc = false;
    while (n_open_ != 0) {
      int b = 0;
      while (bq_[b].empty()) ++b;
      if (gen_left > 0 && b < base_bucket) {
        // excursion below the bucket whose generation is unfinished
        if (!in_exc) { in_exc = true; exc_pops_this_root = 0; }
        ++exc_pops_this_root;
        ++st.excursion_pops;
        ++st.exc_by_bucket[base_bucket & 31];
      } else {
        if (in_exc) { { int lb = 0; while ((1ull << (lb + 1)) <= exc_pops_this_root) ++lb; st.exc_size_hist[lb]++; st.exc_pops_hist[lb] += exc_pops_this_root; } in_exc = false; }
        if (gen_left == 0 || b != base_bucket) {
          base_bucket = b;
          gen_left = bq_[b].size();
          st.gen_sizes.push_back(gen_left);
          ++st.generations;
        }
        --gen_left;
        ++st.base_by_bucket[b & 31];
      }
      const LIdx3 g = bq_[b].front();
      bq_[b].pop_front();
      --n_open_;
      ++st.pops;
      EsdfVoxel* voxel = esdf_layer_->getVoxelPtrByGlobalIndex(g);
      voxel->in_queue = false;
      if (!voxel->observed || voxel->distance >= config_.max_distance_m || voxel->distance <= -config_.max_distance_m) {
        ++st.skipped;
        continue;
      }
      if (bucketOf(voxel->distance) < b) ++st.stale;
      if (hazardous(g, *voxel)) ++st.hazardous;
      for (int idx = 0; idx < 26; ++idx) {
        const Idx3 d = NeighborhoodLut::offset(idx);
        const LIdx3 n{g.x + d.x, g.y + d.y, g.z + d.z};
        const float distance = NeighborhoodLut::distance(idx) * voxel_size_;
        EsdfVoxel* nv = esdf_layer_->getVoxelPtrByGlobalIndex(n);
        if (nv == nullptr || !nv->observed || nv->fixed) continue;
        const Idx3 new_parent{-d.x, -d.y, -d.z};
        bool upd = false;
        float nd = 0;
        if (voxel->distance > 0 && nv->distance > 0) {
          if (voxel->distance + distance + config_.min_diff_m < nv->distance) { nd = voxel->distance + distance; upd = true; }
        } else if (voxel->distance <= 0 && nv->distance <= 0) {
          if (voxel->distance - distance - config_.min_diff_m > nv->distance) { nd = voxel->distance - distance; upd = true; }
        } else {
          const float potential = voxel->distance - signum(voxel->distance) * distance;
          if (std::abs(potential - nv->distance) > distance) {
            if (static_cast<float>(signum(potential)) == nv->distance) nd = potential;
            else nd = signum(nv->distance) * distance;
            upd = true;
          }
        }
        if (upd) {
          ++st.relax;
          if (seq_watch && n == seq_watch_g) std::fprintf(stderr, "[seq] pop #%llu (%lld %lld %lld) d=%.9g bucket %d -> watch %.9g -> %.9g lut %d inq=%d\n", (unsigned long long)st.pops, (long long)g.x, (long long)g.y, (long long)g.z, voxel->distance, b, nv->distance, nd, idx, (int)nv->in_queue);
          nv->distance = nd;
          nv->parent = new_parent;
          if (config_.multi_queue || !nv->in_queue) {
            const int nb = bucketOf(nd);
            if (nb < b) ++st.lower_pushes;
            if (nb == b) ++st.same_bucket_pushes;
            qpush(n, nd);
            nv->in_queue = true;
          }
        }
      }
    }
    if (in_exc) { int lb = 0; while ((1ull << (lb + 1)) <= exc_pops_this_root) ++lb; st.exc_size_hist[lb]++; st.exc_pops_hist[lb] += exc_pops_this_root; }
  }
};

struct Model {
  Model(float vs) : tsdf(vs, 16), esdf(vs, 16), esdf2(vs, 16) {}
  Layer<TsdfVoxel> tsdf;
  Layer<EsdfVoxel> esdf, esdf2;
  std::unique_ptr<FastTsdfIntegrator> fast;
  std::unique_ptr<ModelEsdf> e, e2;
};

}  // namespace

extern "C" {
void* eom_create(float voxel) {
  auto* m = new Model(voxel);
  TsdfConfig c;
  c.default_truncation_distance = 4 * voxel;
  c.integrator_threads = std::getenv("EOM_THREADS") ? std::atoi(std::getenv("EOM_THREADS")) : 8;
  m->fast.reset(new FastTsdfIntegrator(c, &m->tsdf));
  EsdfConfig ec;
  ec.min_distance_m = 2 * voxel;
  if (std::getenv("EOM_MULTI_QUEUE")) ec.multi_queue = std::atoi(std::getenv("EOM_MULTI_QUEUE")) != 0;
  if (std::getenv("EOM_BUCKETS")) ec.num_buckets = std::atoi(std::getenv("EOM_BUCKETS"));
  if (std::getenv("EOM_SPHERES")) {   // "clear,occupied" radii in metres
    float c = 0, o = 0;
    std::sscanf(std::getenv("EOM_SPHERES"), "%f,%f", &c, &o);
    ec.clear_sphere_radius = c;
    ec.occupied_sphere_radius = o;
  }
  m->e.reset(new ModelEsdf(ec, &m->tsdf, &m->esdf));
  m->e2.reset(new ModelEsdf(ec, &m->tsdf, &m->esdf2));
  m->e2->mode = 1;
  return m;
}
void eom_robot(void* h, const float* pos) {
  auto* m = static_cast<Model*>(h);
  m->e->addNewRobotPosition(Vec3f{pos[0], pos[1], pos[2]});
  m->e2->addNewRobotPosition(Vec3f{pos[0], pos[1], pos[2]});
}
void eom_integrate(void* h, const float* pos, const float* q, const float* pts, const uint8_t* rgba, size_t n) {
  auto* m = static_cast<Model*>(h);
  fastResetCounter() = 0;
  Transformation T;
  T.t = {pos[0], pos[1], pos[2]};
  T.qw = q[0]; T.qx = q[1]; T.qy = q[2]; T.qz = q[3];
  m->fast->integratePointCloud(T, reinterpret_cast<const Vec3f*>(pts), reinterpret_cast<const Color*>(rgba), n, false);
}
void eom_update(void* h, int verbose) {
  auto* m = static_cast<Model*>(h);
  m->e->st = Stats();
  m->e->update(false);
  const Stats& s = m->e->st;
  std::printf("pops %llu skipped %llu raise_pops %llu relax %llu pushes %llu same_bucket %llu | stale %llu hazardous %llu | "
              "lower_pushes %llu excursion_pops %llu | generations %llu\n",
              (unsigned long long)s.pops, (unsigned long long)s.skipped, (unsigned long long)s.raise_pops,
              (unsigned long long)s.relax, (unsigned long long)s.pushes, (unsigned long long)s.same_bucket_pushes,
              (unsigned long long)s.stale, (unsigned long long)s.hazardous, (unsigned long long)s.lower_pushes,
              (unsigned long long)s.excursion_pops, (unsigned long long)s.generations);
  if (verbose) {
    std::printf("  per base bucket base/exc pops:");
    for (int k = 0; k < 20; ++k) std::printf(" %d:%llu/%llu", k, (unsigned long long)s.base_by_bucket[k], (unsigned long long)s.exc_by_bucket[k]);
    std::printf("\n");
    std::printf("  gen sizes:");
    for (uint64_t g : s.gen_sizes) std::printf(" %llu", (unsigned long long)g);
    std::printf("\n  excursion size hist:");
    for (auto& kv : s.exc_size_hist) std::printf(" 2^%d:%llu(%llu)", kv.first, (unsigned long long)kv.second, (unsigned long long)s.exc_pops_hist.at(kv.first));
    std::printf("\n");
  }
}
// the batched replay on the second ESDF layer, compared with the sequential one voxel by voxel
void eom_check_counts(void* h, unsigned long long* out) {
  auto* m = static_cast<Model*>(h);
  out[0] = m->e2->emul_check_runs;
  out[1] = m->e2->emul_check_failures;
}
void eom_set_mode(void* h, int mode) { static_cast<Model*>(h)->e2->mode = mode; }
long eom_update_parallel(void* h, size_t kmax, size_t smax, int max_iters) {
  auto* m = static_cast<Model*>(h);
  m->e2->st = Stats();
  m->e2->ps = ModelEsdf::ParStats();
  m->e2->kmax = kmax; m->e2->smax = smax; m->e2->max_iters = max_iters;
  m->e2->update(true);
  long diff = 0, n = 0;
  for (auto& kv : m->esdf.block_map) {
    auto it = m->esdf2.block_map.find(kv.first);
    if (it == m->esdf2.block_map.end()) { diff += 4096; continue; }
    for (size_t i = 0; i < kv.second->num_voxels; ++i) {
      const EsdfVoxel& a = kv.second->voxels[i];
      const EsdfVoxel& c = it->second->voxels[i];
      ++n;
      if (std::memcmp(&a.distance, &c.distance, 4) != 0 || a.observed != c.observed || a.in_queue != c.in_queue ||
          a.fixed != c.fixed || !(a.parent == c.parent) || a.hallucinated != c.hallucinated) {
        ++diff;
        if (std::getenv("EOM_DUMP") && diff <= 20)
          std::printf("    diff blk (%d %d %d) lin %zu: seq d=%.9g q=%d par=(%d %d %d) fixed=%d | par d=%.9g q=%d par=(%d %d %d)\n", kv.first.x, kv.first.y, kv.first.z, i,
                      a.distance, a.in_queue, a.parent.x, a.parent.y, a.parent.z, a.fixed, c.distance, c.in_queue, c.parent.x, c.parent.y, c.parent.z);
      }
    }
  }
  if (m->e2->mode == 2) {
    const rp::Ctl& c = m->e2->emul_ctl;
    std::printf("  emul: raise pops %llu in %llu steps | pops %llu relax %llu supersteps %llu iters %llu folds %llu exc %llu cuts(iters %llu smax %llu) steps %llu error %u | voxels %ld DIFF %ld\n",
                c.st_raise_pops, c.st_raise_steps, c.st_pops, c.st_relax, c.st_supersteps, c.st_iters, c.st_folds, c.st_exc, c.st_cut_iters, c.st_cut_smax, c.st_steps, c.error, n, diff);
    static const char* names[] = {"done", "begin", "place", "fold", "apply", "sim", "mincut", "cfold", "rank", "rwrite", "push", "cleanup", "raise"};
    std::printf("  poison %llu trunc_q %llu trunc_rank %llu retries %llu\n", c.st_poison, c.st_trunc_q, c.st_trunc_rank, c.st_retries);
    std::printf("  steps:");
    for (int k = 1; k < 13; ++k) std::printf(" %s %llu(%llu)", names[k], c.st_phase_steps[k], c.st_phase_threads[k]);
    std::printf("\n");
    return diff;
  }
  const auto& p = m->e2->ps;
  std::printf("  parallel: pops %llu  supersteps %llu (small %llu, iters in small %llu) iterations %llu folds %llu exc_records %llu cuts(iters %llu, smax %llu) max_iters %llu max_events %llu | voxels %ld DIFF %ld\n",
              (unsigned long long)m->e2->st.pops, (unsigned long long)p.supersteps, (unsigned long long)p.small_steps,
              (unsigned long long)p.small_iters, (unsigned long long)p.iterations,
              (unsigned long long)p.folds, (unsigned long long)p.exc_records, (unsigned long long)p.cut_iters,
              (unsigned long long)p.cut_smax, (unsigned long long)p.max_iters_seen, (unsigned long long)p.max_events, n, diff);
  std::printf("  iters hist:");
  for (auto& kv : p.iters_hist) std::printf(" %d:%llu", kv.first, (unsigned long long)kv.second);
  std::printf("\n");
  return diff;
}
}
