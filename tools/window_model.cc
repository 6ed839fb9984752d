// window_model.cc — CPU model of the WINDOWED observed-set replay (design tool, not product code).
//
// The reference walks the kept rays one after the other through ONE lossy set (ApproxHashSet<20,10000>,
// approx_hash_array.h:125-134).  The windowed replay takes W consecutive rays at a time: every ray of the
// window is evaluated against the set content the window found ("frozen") plus the probes of the window's
// earlier rays under the current guess of their probe counts; rays whose count moved are re-evaluated; the
// prefix of the window in front of the first ray that still moved is final and committed, the next window
// starts behind it.  This model counts windows, iterations and shared-slot probes for a frame so that the
// GPU kernel (k_fast_window) can be sized before spending GPU time, and checks the result against the
// sequential walk.
//
//   g++ -O2 -std=c++17 -shared -fPIC -I oracle tools/window_model.cc -o tools/libwindow_model.so
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "vbx_tsdf.hpp"

using namespace orc;

namespace {
struct Probe {
  uint32_t slot, time, hash;  // time = (ray - r0) << 10 | k   (k < 1024)
};
}  // namespace

extern "C" int window_model(const float* pos, const float* quat, const float* pts, uint32_t n, float voxel, int wmax,
                            int pmax, int kmax, int verbose) {
  TsdfConfig cfg;
  cfg.default_truncation_distance = 4 * voxel;
  const float voxel_size_inv = 1.0 / voxel;
  Transformation T;
  T.t = {pos[0], pos[1], pos[2]};
  T.qw = quat[0]; T.qx = quat[1]; T.qy = quat[2]; T.qz = quat[3];
  const Vec3f origin = T.getPosition();
  ApproxHashSet<20, 10000> start_set;
  start_set.resetApproxSet();
  std::vector<uint32_t> off{0}, hash;
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
    while (rc.nextRayIndex(&g)) hash.push_back((uint32_t)longIndexHash(g));
    off.push_back((uint32_t)hash.size());
  }
  const uint32_t R = (uint32_t)off.size() - 1;
  const int maxc = cfg.max_consecutive_ray_collisions;
  std::vector<uint32_t> Tseq(R);
  uint64_t sum_seq = 0;
  {
    ApproxHashSet<20, 10000> obs;
    obs.resetApproxSet();
    for (uint32_t r = 0; r < R; ++r) {
      int cons = 0;
      uint32_t t = 0;
      for (uint32_t k = off[r]; k < off[r + 1]; ++k) {
        ++t;
        if (!obs.replaceHash(hash[k])) ++cons; else cons = 0;
        if (cons > maxc) break;
      }
      Tseq[r] = t;
      sum_seq += t;
    }
  }
  // windowed replay.  set content: 0xFFFFFFFF'FFFFFFFF never matches; the fresh set of a frame holds 0 in every
  // slot (hash 0 reads as present) except that we model the content as u64 to keep "empty" apart
  std::vector<uint64_t> set(1u << 20, 0ull);
  std::vector<uint32_t> Tw(R, 0);
  std::vector<Probe> P;
  uint64_t windows = 0, iters = 0, shared = 0, live_total = 0, max_live = 0, cut_short = 0, evals = 0;
  uint64_t hist_it[8] = {0};
  std::vector<uint32_t> Tg, Tn;
  uint32_t r0 = 0;
  while (r0 < R) {
    ++windows;
    // window extent: at most wmax rays; probe budget judged by the frozen evaluation below
    uint32_t r1 = std::min<uint32_t>(R, r0 + (uint32_t)wmax);
    const uint32_t W = r1 - r0;
    Tg.assign(W, 0);
    Tn.assign(W, 0);
    // iteration 0: frozen evaluation (a ray's own earlier probes count as predecessors)
    auto eval_ray = [&](uint32_t j, bool use_table) -> uint32_t {
      const uint32_t r = r0 + j;
      const uint32_t len = off[r + 1] - off[r];
      int cons = 0;
      for (uint32_t k = 0; k < len; ++k) {
        ++evals;
        const uint32_t h = hash[off[r] + k];
        const uint32_t slot = h & 0xFFFFFu;
        uint64_t content = set[slot];
        const uint32_t mytime = (j << 10) | k;
        if (use_table) {
          // latest live probe of the window with this slot and time < mytime
          auto it = std::lower_bound(P.begin(), P.end(), Probe{slot, mytime, 0}, [](const Probe& a, const Probe& b) {
            return a.slot != b.slot ? a.slot < b.slot : a.time < b.time;
          });
          if (it != P.begin()) {
            --it;
            if (it->slot == slot) content = it->hash;
          }
        } else {
          // own earlier probes only
          for (uint32_t k2 = k; k2-- > 0;) {
            const uint32_t h2 = hash[off[r] + k2];
            if ((h2 & 0xFFFFFu) == slot) { content = h2; break; }
          }
        }
        if (content == (uint64_t)h) ++cons; else cons = 0;
        if (cons > maxc) return k + 1;
      }
      return len;
    };
    uint64_t budget = 0;
    for (uint32_t j = 0; j < W; ++j) {
      Tg[j] = eval_ray(j, false);
      budget += Tg[j];
      if (pmax > 0 && budget > (uint64_t)pmax && j + 1 < W) {  // window ends where the probe budget is used up
        r1 = r0 + j + 1;
        break;
      }
    }
    const uint32_t Wn = r1 - r0;
    uint32_t c = Wn;  // first ray that moved in the last iteration
    int it = 0;
    for (;;) {
      ++it;
      ++iters;
      // table of the live probes under the guess
      P.clear();
      for (uint32_t j = 0; j < Wn; ++j)
        for (uint32_t k = 0; k < Tg[j]; ++k) {
          const uint32_t h = hash[off[r0 + j] + k];
          P.push_back(Probe{h & 0xFFFFFu, (j << 10) | k, h});
        }
      std::sort(P.begin(), P.end(), [](const Probe& a, const Probe& b) { return a.slot != b.slot ? a.slot < b.slot : a.time < b.time; });
      if (it == 1) {
        live_total += P.size();
        max_live = std::max<uint64_t>(max_live, P.size());
        for (size_t i = 1; i < P.size(); ++i) shared += (P[i].slot == P[i - 1].slot);
      }
      c = Wn;
      for (uint32_t j = 0; j < Wn; ++j) {
        Tn[j] = eval_ray(j, true);
        if (Tn[j] != Tg[j] && c == Wn) c = j;
      }
      Tg.swap(Tn);
      Tg.resize(Wn);
      Tn.resize(Wn);
      if (c == Wn) break;           // nothing moved: the whole window is final
      if (it >= kmax && c > 0) break;  // commit the stable prefix
    }
    hist_it[std::min(it, 7)]++;
    if (c < Wn) ++cut_short;
    // commit rays [r0, r0 + c): their probes in time order
    // (Tg was swapped: for j < c Tn == Tg, either holds the final count)
    for (uint32_t j = 0; j < c; ++j) {
      const uint32_t r = r0 + j;
      Tw[r] = Tg[j];
      for (uint32_t k = 0; k < Tg[j]; ++k) set[hash[off[r] + k] & 0xFFFFFu] = hash[off[r] + k];
    }
    if (verbose > 1) fprintf(stderr, "window %llu: rays [%u,%u) committed %u after %d iterations\n", (unsigned long long)windows, r0, r1, c, it);
    r0 += c;
  }
  uint32_t bad = 0;
  for (uint32_t r = 0; r < R; ++r) bad += (Tw[r] != Tseq[r]);
  printf("{\"R\": %u, \"probes_seq\": %llu, \"mismatch\": %u, \"wmax\": %d, \"pmax\": %d, \"kmax\": %d, \"windows\": %llu, \"iterations\": %llu, "
         "\"cut_short\": %llu, \"live_per_window\": %.0f, \"max_live\": %llu, \"shared_frac\": %.4f, \"evals\": %llu, "
         "\"iter_hist\": [%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu]}\n",
         R, (unsigned long long)sum_seq, bad, wmax, pmax, kmax, (unsigned long long)windows, (unsigned long long)iters,
         (unsigned long long)cut_short, (double)live_total / (double)windows, (unsigned long long)max_live,
         (double)shared / (double)std::max<uint64_t>(1, live_total), (unsigned long long)evals,
         (unsigned long long)hist_it[0], (unsigned long long)hist_it[1], (unsigned long long)hist_it[2], (unsigned long long)hist_it[3],
         (unsigned long long)hist_it[4], (unsigned long long)hist_it[5], (unsigned long long)hist_it[6], (unsigned long long)hist_it[7]);
  fflush(stdout);
  return bad ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------
// frontier_model: the sort-based Jacobi rounds of the product path (k_strict_keys / sort / outcome / scan),
// restricted to a sliding horizon behind the settled frontier F: a round evaluates the rays [F, B) whose
// guessed probes add up to ~pbudget, the rays in front of the first one that moved are final and committed,
// and the next round starts there.  Counts rounds and evaluated probes for a frame.
//   guess_mode 0: exact-set solution for every ray (what the product path starts from)
//   guess_mode 1: min(len, guess_const)
extern "C" int frontier_model(const float* pos, const float* quat, const float* pts, uint32_t n, float voxel, int pbudget,
                              int guess_mode, int guess_const, int grow_mult, int verbose) {
  TsdfConfig cfg;
  cfg.default_truncation_distance = 4 * voxel;
  const float voxel_size_inv = 1.0 / voxel;
  Transformation T;
  T.t = {pos[0], pos[1], pos[2]};
  T.qw = quat[0]; T.qx = quat[1]; T.qy = quat[2]; T.qz = quat[3];
  const Vec3f origin = T.getPosition();
  ApproxHashSet<20, 10000> start_set;
  start_set.resetApproxSet();
  std::vector<uint32_t> off{0}, hash;
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
    while (rc.nextRayIndex(&g)) hash.push_back((uint32_t)longIndexHash(g));
    off.push_back((uint32_t)hash.size());
  }
  const uint32_t R = (uint32_t)off.size() - 1;
  const int maxc = cfg.max_consecutive_ray_collisions;
  std::vector<uint32_t> Tseq(R), Tg(R);
  {
    ApproxHashSet<20, 10000> obs;
    obs.resetApproxSet();
    std::vector<uint8_t> seen;  // exact set over hashes (injective for a frame's voxels for all practical purposes)
    std::vector<uint32_t> ex_keys;
    for (uint32_t r = 0; r < R; ++r) {
      int cons = 0;
      uint32_t t = 0;
      for (uint32_t k = off[r]; k < off[r + 1]; ++k) {
        ++t;
        if (!obs.replaceHash(hash[k])) ++cons; else cons = 0;
        if (cons > maxc) break;
      }
      Tseq[r] = t;
    }
    // exact-set guess
    std::vector<uint32_t> sorted_all;
    {
      // open-addressing exact set keyed by the 32-bit hash
      const size_t cap = 1u << 27;
      std::vector<uint32_t> tab(cap, 0xFFFFFFFFu);
      auto insert = [&](uint32_t h) -> bool {
        size_t i = (h * 2654435761u) & (cap - 1);
        while (tab[i] != 0xFFFFFFFFu) {
          if (tab[i] == h) return false;
          i = (i + 1) & (cap - 1);
        }
        tab[i] = h;
        return true;
      };
      for (uint32_t r = 0; r < R; ++r) {
        int cons = 0;
        uint32_t t = 0;
        for (uint32_t k = off[r]; k < off[r + 1]; ++k) {
          ++t;
          if (!insert(hash[k])) ++cons; else cons = 0;
          if (cons > maxc) break;
        }
        const uint32_t len = off[r + 1] - off[r];
        Tg[r] = guess_mode == 0 ? t : std::min<uint32_t>(len, (uint32_t)guess_const);
      }
    }
    if (guess_mode == 2) {
      // guess = the rays taken guess_const at a time: every ray of a window evaluated against the set content the
      // window found (no look at the other rays of the window), then the window's probes written in time order
      std::vector<uint64_t> gset(1u << 20, 0ull);
      uint64_t wrong = 0;
      for (uint32_t r0 = 0; r0 < R; r0 += (uint32_t)guess_const) {
        const uint32_t r1 = std::min<uint32_t>(R, r0 + (uint32_t)guess_const);
        for (uint32_t r = r0; r < r1; ++r) {
          int cons = 0;
          uint32_t t = 0;
          for (uint32_t k = off[r]; k < off[r + 1]; ++k) {
            ++t;
            if (gset[hash[k] & 0xFFFFFu] == (uint64_t)hash[k]) ++cons; else cons = 0;
            if (cons > maxc) break;
          }
          Tg[r] = t;
          wrong += (t != Tseq[r]);
        }
        for (uint32_t r = r0; r < r1; ++r)
          for (uint32_t k = 0; k < Tg[r]; ++k) gset[hash[off[r] + k] & 0xFFFFFu] = hash[off[r] + k];
      }
      fprintf(stderr, "windowed guess (W=%d): %llu of %u rays differ from the sequential result\n", guess_const,
              (unsigned long long)wrong, R);
    }
  }
  std::vector<uint64_t> set(1u << 20, 0ull);
  struct E { uint32_t slot, p, hash; };
  std::vector<E> K;
  std::vector<uint8_t> col;
  uint64_t rounds = 0, evaluated = 0;
  uint32_t F = 0;
  std::vector<uint32_t> poff;
  while (F < R) {
    // horizon
    uint32_t B = F;
    uint64_t acc = 0;
    while (B < R && (B == F || acc + Tg[B] <= (uint64_t)pbudget)) acc += Tg[B++];
    ++rounds;
    evaluated += acc;
    K.clear();
    poff.assign(B - F + 1, 0);
    for (uint32_t r = F; r < B; ++r) {
      poff[r - F + 1] = poff[r - F] + Tg[r];
      for (uint32_t k = 0; k < Tg[r]; ++k) {
        const uint32_t h = hash[off[r] + k];
        K.push_back(E{h & 0xFFFFFu, poff[r - F] + k, h});
      }
    }
    std::stable_sort(K.begin(), K.end(), [](const E& a, const E& b) { return a.slot < b.slot; });
    col.assign(K.size(), 0);
    for (size_t i = 0; i < K.size(); ++i) {
      uint64_t content = (i > 0 && K[i - 1].slot == K[i].slot) ? (uint64_t)K[i - 1].hash : set[K[i].slot];
      col[K[i].p] = content == (uint64_t)K[i].hash;
    }
    uint32_t c = B;
    for (uint32_t r = F; r < B; ++r) {
      const uint32_t len = off[r + 1] - off[r];
      const uint32_t t = Tg[r];
      int cons = 0;
      uint32_t tn = t;
      bool broke = false;
      for (uint32_t k = 0; k < t; ++k) {
        cons = col[poff[r - F] + k] ? cons + 1 : 0;
        if (cons > maxc) { tn = k + 1; broke = true; break; }
      }
      if (!broke && t < len) tn = std::min(len, std::max((uint32_t)grow_mult * t, t + 16u));
      if (tn != t && c == B) c = r;
      Tg[r] = tn;
    }
    // commit [F, c)
    for (uint32_t r = F; r < c; ++r)
      for (uint32_t k = 0; k < Tg[r]; ++k) set[hash[off[r] + k] & 0xFFFFFu] = hash[off[r] + k];
    if (verbose > 1 || (verbose && rounds % 20 == 0)) fprintf(stderr, "round %llu: [%u,%u) %llu probes, first moved %u\n", (unsigned long long)rounds, F, B, (unsigned long long)acc, c);
    if (c == F) {
      // the first ray itself moved: it only depends on committed state and on itself, so it settles within a few rounds
    }
    F = c;
  }
  uint32_t bad = 0;
  uint64_t sum_seq = 0;
  for (uint32_t r = 0; r < R; ++r) { bad += (Tg[r] != Tseq[r]); sum_seq += Tseq[r]; }
  printf("{\"R\": %u, \"probes_seq\": %llu, \"mismatch\": %u, \"pbudget\": %d, \"guess_mode\": %d, \"grow\": %d, \"rounds\": %llu, \"evaluated\": %llu}\n", R,
         (unsigned long long)sum_seq, bad, pbudget, guess_mode, grow_mult, (unsigned long long)rounds, (unsigned long long)evaluated);
  fflush(stdout);
  return bad ? 1 : 0;
}
