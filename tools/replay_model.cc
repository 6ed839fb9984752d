// replay_model.cc — CPU model of the event-driven observed-set replay (design tool, not product code).
//
// Builds the Fast integrator's kept rays and their full surface->sensor voxel-hash lists with the
// ORACLE's restated reference code (isPointValid, start-voxel ApproxHashSet, RayCaster), runs
//   (1) the reference's sequential ApproxHashSet walk          -> T_seq (probes per ray)
//   (2) the same with an exact set                              -> T_exact (the solver's guess)
//   (3) the event-driven fixed-point iteration modelled in plain loops, round by round
// checks (3) == (1) and prints per-round event statistics (changed rays, toggled probes, walk
// lengths over dead candidates, rebases).  Used to size the GPU implementation
// (voxblox_amd/csrc/vbx_kernels_replay.hpp) before spending GPU time.
//
//   g++ -O2 -std=c++17 -shared -fPIC -I oracle tools/replay_model.cc -o tools/libreplay_model.so
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <unordered_set>
#include <vector>

#include "vbx_tsdf.hpp"

using namespace orc;

namespace {
struct Lists {
  std::vector<uint32_t> off;   // R+1
  std::vector<uint32_t> hash;  // per list entry: LongIndexHash truncated to 32 bit
};

struct Entry {  // one candidate probe in the slot-sorted index
  uint32_t slot, r, k, hash;
};

struct Stats {
  uint64_t rounds = 0, changes = 0, events = 0, walk_back = 0, walk_fwd = 0, rebases = 0, cand = 0, rescans = 0;
  uint64_t max_walk = 0;
};

// derive a ray's probe count from the outcomes of its first t probes (tsdf_integrator.cc:531-551)
inline uint32_t derive(const uint8_t* col, uint32_t t, uint32_t cap, uint32_t len, int max_consecutive, bool* broke,
                       bool* overflow) {
  int cons = 0;
  *broke = false;
  *overflow = false;
  for (uint32_t k = 0; k < t; ++k) {
    cons = col[k] ? cons + 1 : 0;
    if (cons > max_consecutive) {
      *broke = true;
      return k + 1;
    }
  }
  if (t >= len) return len;
  uint32_t tn = std::min(len, std::max(4u * t, t + 16u));
  if (tn > cap) {
    tn = cap;
    if (cap == t) *overflow = true;
  }
  return tn;
}
}  // namespace

extern "C" int model_run(const float* pos, const float* quat, const float* pts, uint32_t n, float voxel, int guess_mode,
                         int guess_const, int verbose, int c_mult, int c_add, int g_mult, int g_add, int early_rebase,
                         uint32_t* out_T_by_point, const uint32_t* guess_by_point) {
  TsdfConfig cfg;
  cfg.default_truncation_distance = 4 * voxel;
  const float voxel_size_inv = 1.0 / voxel;
  Transformation T;
  T.t = {pos[0], pos[1], pos[2]};
  T.qw = quat[0]; T.qx = quat[1]; T.qy = quat[2]; T.qz = quat[3];
  const Vec3f origin = T.getPosition();
  ApproxHashSet<20, 10000> start_set;
  start_set.resetApproxSet();
  Lists L;
  std::vector<uint32_t> point_of_ray;
  L.off.push_back(0);
  for (size_t s = 0; s < n; ++s) {
    const size_t i = mixedIndex(s, n);
    const Vec3f pc{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    const float d = norm(pc);
    bool clearing;
    if (d < cfg.min_ray_length_m) continue;
    if (d > cfg.max_ray_length_m) {
      if (!cfg.allow_clear) continue;
      clearing = true;
    } else {
      clearing = false;
    }
    const Vec3f pg = T * pc;
    LIdx3 g = gridIndexFromPointL(pg, cfg.start_voxel_subsampling_factor * voxel_size_inv);
    if (!start_set.replaceHash(longIndexHash(g))) continue;
    RayCaster rc(origin, pg, clearing, cfg.voxel_carving_enabled, cfg.max_ray_length_m, voxel_size_inv,
                 cfg.default_truncation_distance, false);
    while (rc.nextRayIndex(&g)) L.hash.push_back((uint32_t)longIndexHash(g));
    L.off.push_back((uint32_t)L.hash.size());
    point_of_ray.push_back((uint32_t)i);
  }
  const uint32_t R = (uint32_t)L.off.size() - 1;
  const int maxc = cfg.max_consecutive_ray_collisions;
  // (1) sequential reference
  std::vector<uint32_t> Tseq(R), Texact(R);
  {
    ApproxHashSet<20, 10000> obs;
    obs.resetApproxSet();
    for (uint32_t r = 0; r < R; ++r) {
      int cons = 0;
      uint32_t t = 0;
      for (uint32_t k = L.off[r]; k < L.off[r + 1]; ++k) {
        ++t;
        if (!obs.replaceHash(L.hash[k])) ++cons; else cons = 0;
        if (cons > maxc) break;
      }
      Tseq[r] = t;
    }
    std::unordered_set<uint32_t> ex;  // hashes are injective over a frame's voxels for all practical purposes
    for (uint32_t r = 0; r < R; ++r) {
      int cons = 0;
      uint32_t t = 0;
      for (uint32_t k = L.off[r]; k < L.off[r + 1]; ++k) {
        ++t;
        if (!ex.insert(L.hash[k]).second) ++cons; else cons = 0;
        if (cons > maxc) break;
      }
      Texact[r] = t;
    }
  }
  if (out_T_by_point) {
    std::memset(out_T_by_point, 0, (size_t)n * 4);
    for (uint32_t r = 0; r < R; ++r) out_T_by_point[point_of_ray[r]] = Tseq[r];
  }
  uint64_t sum_seq = 0, sum_ex = 0;
  for (uint32_t r = 0; r < R; ++r) { sum_seq += Tseq[r]; sum_ex += Texact[r]; }
  if (verbose) fprintf(stderr, "R=%u list entries=%zu  probes: sequential %llu, exact-set guess %llu\n", R, L.hash.size(),
                       (unsigned long long)sum_seq, (unsigned long long)sum_ex);

  if (verbose > 2) {  // distribution of the final probe counts against the guess and the path length
    uint64_t h_ratio[8] = {0}, h_len[6] = {0};
    for (uint32_t r = 0; r < R; ++r) {
      const uint32_t len = L.off[r + 1] - L.off[r];
      const double q = (double)Tseq[r] / std::max(1u, Texact[r]);
      h_ratio[q <= 1 ? 0 : q <= 1.5 ? 1 : q <= 2 ? 2 : q <= 3 ? 3 : q <= 4 ? 4 : q <= 8 ? 5 : q <= 16 ? 6 : 7]++;
      const double f = (double)Tseq[r] / std::max(1u, len);
      h_len[f < 0.1 ? 0 : f < 0.25 ? 1 : f < 0.5 ? 2 : f < 0.9 ? 3 : f < 1 ? 4 : 5]++;
    }
    fprintf(stderr, "T_seq/T_exact: <=1 %llu, <=1.5 %llu, <=2 %llu, <=3 %llu, <=4 %llu, <=8 %llu, <=16 %llu, more %llu\n",
            (unsigned long long)h_ratio[0], (unsigned long long)h_ratio[1], (unsigned long long)h_ratio[2], (unsigned long long)h_ratio[3],
            (unsigned long long)h_ratio[4], (unsigned long long)h_ratio[5], (unsigned long long)h_ratio[6], (unsigned long long)h_ratio[7]);
    fprintf(stderr, "T_seq/len: <0.1 %llu, <0.25 %llu, <0.5 %llu, <0.9 %llu, <1 %llu, =1 %llu\n", (unsigned long long)h_len[0],
            (unsigned long long)h_len[1], (unsigned long long)h_len[2], (unsigned long long)h_len[3], (unsigned long long)h_len[4],
            (unsigned long long)h_len[5]);
    return 0;
  }
  // (3) event-driven model
  std::vector<uint32_t> Tc(R), C(R), coff(R + 1);
  for (uint32_t r = 0; r < R; ++r) {
    const uint32_t len = L.off[r + 1] - L.off[r];
    Tc[r] = guess_mode == 0 ? Texact[r] : std::min<uint32_t>(len, (uint32_t)guess_const);
    if (guess_mode == 2) Tc[r] = (guess_by_point && guess_by_point[point_of_ray[r]]) ? std::min(len, guess_by_point[point_of_ray[r]]) : Texact[r];
    if (guess_mode == 3) Tc[r] = (guess_by_point && guess_by_point[point_of_ray[r]]) ? std::min(len, std::max(Texact[r], guess_by_point[point_of_ray[r]])) : Texact[r];
  }
  Stats st;
  std::vector<Entry> K;
  std::vector<uint32_t> posOf;
  std::vector<uint8_t> col;
  std::vector<uint8_t> grow_more(R, 0);
  struct Change { uint32_t r, t_old, t_new; };
  std::vector<Change> cl, cl_next;
  std::vector<uint32_t> dirty;
  std::vector<uint8_t> dflag(R, 0);
  auto live = [&](const Entry& e) { return e.k < Tc[e.r]; };
  auto eval = [&](uint32_t i) -> uint8_t {  // outcome of the live candidate at sorted position i
    const Entry& e = K[i];
    uint64_t w = 0;
    for (uint32_t j = i; j-- > 0;) {
      if (K[j].slot != e.slot) break;
      ++w;
      if (live(K[j])) {
        st.walk_back += w;
        st.max_walk = std::max(st.max_walk, w);
        return K[j].hash == e.hash;
      }
    }
    st.walk_back += w;
    st.max_walk = std::max(st.max_walk, w);
    return 0;  // fresh set content (offset moved on since the last frame): nothing there
  };
  for (;;) {
    // ---- rebase: candidates from the current guess
    ++st.rebases;
    for (uint32_t r = 0; r < R; ++r) {
      const uint32_t len = L.off[r + 1] - L.off[r];
      const uint32_t want = grow_more[r] ? (uint32_t)g_mult * Tc[r] + (uint32_t)g_add : std::max((uint32_t)c_mult * Tc[r], Tc[r] + (uint32_t)c_add);
      C[r] = std::min(len, want);
      grow_more[r] = 0;
    }
    coff[0] = 0;
    for (uint32_t r = 0; r < R; ++r) coff[r + 1] = coff[r] + C[r];
    const uint32_t Pc = coff[R];
    st.cand += Pc;
    K.resize(Pc);
    for (uint32_t r = 0; r < R; ++r)
      for (uint32_t k = 0; k < C[r]; ++k) {
        const uint32_t h = L.hash[L.off[r] + k];
        K[coff[r] + k] = Entry{h & 0xFFFFFu, r, k, h};
      }
    std::stable_sort(K.begin(), K.end(), [](const Entry& a, const Entry& b) { return a.slot < b.slot; });
    posOf.assign(Pc, 0);
    col.assign(Pc, 0);
    for (uint32_t i = 0; i < Pc; ++i) posOf[coff[K[i].r] + K[i].k] = i;
    for (uint32_t i = 0; i < Pc; ++i)
      if (live(K[i])) col[coff[K[i].r] + K[i].k] = eval(i);
    cl.clear();
    uint32_t n_over = 0;
    for (uint32_t r = 0; r < R; ++r) {
      bool broke, over;
      const uint32_t len = L.off[r + 1] - L.off[r];
      const uint32_t tn = derive(&col[coff[r]], Tc[r], C[r], len, maxc, &broke, &over);
      if (over) { grow_more[r] = 1; ++n_over; }
      if (tn != Tc[r]) { cl.push_back({r, Tc[r], tn}); Tc[r] = tn; }
    }
    if (verbose) fprintf(stderr, "rebase %llu: candidates %u, first change list %zu\n", (unsigned long long)st.rebases, Pc, cl.size());
    // ---- event rounds
    uint32_t round = 0;
    while (!cl.empty()) {
      ++round; ++st.rounds;
      st.changes += cl.size();
      uint64_t ev = 0;
      dirty.clear();
      for (const Change& c : cl) {
        const uint32_t lo = std::min(c.t_old, c.t_new), hi = std::max(c.t_old, c.t_new);
        const bool added = c.t_new > c.t_old;
        if (added && !dflag[c.r]) { dflag[c.r] = 1; dirty.push_back(c.r); }
        for (uint32_t k = lo; k < hi; ++k) {
          ++ev;
          const uint32_t i = posOf[coff[c.r] + k];
          if (added) col[coff[c.r] + k] = eval(i);
          // successor: the next live candidate of the slot
          uint64_t w = 0;
          for (uint32_t j = i + 1; j < Pc && K[j].slot == K[i].slot; ++j) {
            ++w;
            if (live(K[j])) {
              const uint8_t o = eval(j);
              uint8_t& cur = col[coff[K[j].r] + K[j].k];
              if (o != cur) {
                cur = o;
                if (!dflag[K[j].r]) { dflag[K[j].r] = 1; dirty.push_back(K[j].r); }
              }
              break;
            }
          }
          st.walk_fwd += w;
          st.max_walk = std::max(st.max_walk, w);
        }
      }
      st.events += ev;
      st.rescans += dirty.size();
      cl_next.clear();
      for (uint32_t r : dirty) {
        dflag[r] = 0;
        bool broke, over;
        const uint32_t len = L.off[r + 1] - L.off[r];
        const uint32_t tn = derive(&col[coff[r]], Tc[r], C[r], len, maxc, &broke, &over);
        if (over) { if (!grow_more[r]) ++n_over; grow_more[r] = 1; } else if (grow_more[r]) { grow_more[r] = 0; --n_over; }
        if (tn != Tc[r]) { cl_next.push_back({r, Tc[r], tn}); Tc[r] = tn; }
      }
      if (verbose > 1 || (verbose && (round <= 12 || round % 10 == 0)))
        fprintf(stderr, "  round %u: %zu changed rays, %llu probe events, %zu rescans -> %zu changed\n", round, cl.size(),
                (unsigned long long)ev, dirty.size(), cl_next.size());
      cl.swap(cl_next);
      if (early_rebase && n_over >= std::max<size_t>(64, (size_t)early_rebase * cl.size())) break;  // most of the activity is stuck at the caps
    }
    // overflow rays: still unsatisfied at their candidate cap?
    n_over = 0;
    for (uint32_t r = 0; r < R; ++r) n_over += grow_more[r];
    if (verbose) fprintf(stderr, "  converged inside candidates after %u rounds, %u rays at their cap\n", round, n_over);
    if (!n_over && cl.empty()) break;
  }
  uint32_t bad = 0;
  for (uint32_t r = 0; r < R; ++r) bad += (Tc[r] != Tseq[r]);
  printf("{\"R\": %u, \"probes_seq\": %llu, \"probes_guess\": %llu, \"mismatch\": %u, \"rebases\": %llu, \"rounds\": %llu, "
         "\"changes\": %llu, \"events\": %llu, \"rescans\": %llu, \"candidates\": %llu, \"walk_back\": %llu, \"walk_fwd\": %llu, "
         "\"max_walk\": %llu}\n",
         R, (unsigned long long)sum_seq, (unsigned long long)sum_ex, bad, (unsigned long long)st.rebases,
         (unsigned long long)st.rounds, (unsigned long long)st.changes, (unsigned long long)st.events,
         (unsigned long long)st.rescans, (unsigned long long)st.cand, (unsigned long long)st.walk_back,
         (unsigned long long)st.walk_fwd, (unsigned long long)st.max_walk);
  fflush(stdout);
  return bad ? 1 : 0;
}
