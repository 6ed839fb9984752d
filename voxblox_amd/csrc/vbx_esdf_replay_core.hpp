// vbx_esdf_replay_core.hpp — processOpenSet in the REFERENCE'S OWN ORDER, in parallel (cfg.reference_order = 1)
//
// The reference pops one voxel at a time from a BucketQueue (lowest non-empty bucket, FIFO inside) and relaxes its
// 26 neighbours (esdf_integrator.cc:371-496, bucket_queue.h:41-80); with min_diff_m > 0, in_queue de-duplication
// and the sign-mismatch branch the result is DEFINED by that order.  This file replays it exactly, many pops at a
// time:
//
//   super-step   the FIFO prefix (<= kmax entries) of the lowest non-empty bucket b = BASE records; a record is
//                one pop.  A pop whose push lands BELOW b spawns an EXCURSION record: the reference pops that
//                entry (and what it pushes below b, transitively) before it returns to b.  Records carry a pop
//                time T = (base index << 24 | rank inside the base record's excursion); ranks come from running
//                the queue discipline over the excursion's records only (rp_sim_subtree).
//   fold         every voxel next to a record ("target") owns the list of events that concern it — an offer from
//                each record on a neighbour (the reference's `for idx` body for that neighbour), a pop for each
//                record on itself — sorts it by T and replays it from the voxel's state at the start of the
//                super-step, taking each popping voxel's distance / parent AT ITS POP TIME from the previous
//                iteration.  Outputs: the state of its own records at their pop times, which pushes went below b
//                (liveness of excursion records on this voxel, new ones), the voxel's final state.
//   iterate      until nothing changes.  Every event depends only on events with smaller T, so after k iterations
//                the first k events are exact and stay exact: a self-consistent prefix IS the reference's result.
//                If the iteration is stopped early (iteration cap, an excursion larger than smax, capacity), the
//                prefix in front of the earliest change is committed and the rest stays queued.
//   commit       targets write their state as of the cut; the committed records' pushes are appended to the bucket
//                FIFOs in (T, LUT index) order = the order in which the reference pushed them.
//
// WHICH targets an iteration folds: the ones somebody marked dirty — PH_PLACE_BASE's are all new and are folded without a
// list (Cfg::fold_all), PH_APPLY marks those of the records whose state or liveness changed, and a ranking marks those of
// the records whose pop ORDER it changed (Cfg::mark_moved, rp_mark_rec_targets).  tools/esdf_order_model.cc with EOM_CHECK=1
// folds every target once more at every fixed point and must find nothing to change in front of the cut.
//
// The code is a set of PHASES: plain functions of (arguments, thread id) with no synchronisation inside — a phase
// runs to completion before the next starts (on the device a kernel boundary, vbx_kernels_esdf_replay.hpp; in
// tools/esdf_order_model.cc (mode 2) a serial loop, which is how this file is checked against the oracle without a GPU) —
// and rp_control, run by one thread after every phase, which picks the next.  The two SCAN phases (PH_RANK, PH_PUSH) are
// collective: this file gives their per-item count / apply functions, the prefix sum itself lives outside.
//
// The includer defines RP_FN (function qualifiers), RP_INC (returning increment of a counter that every caller in a wave
// shares: the device sends one atomic per wave), RP_LD / RP_LD64 (coherent read of a word other workgroups updated
// with atomics) and provides atomicAdd / atomicCAS / atomicMin / atomicOr / atomicExch on uint32_t and unsigned long long.
// Optional: RP_WG_DIRTY_PUSH(t) -> bool, a per-workgroup collector of dirty marks (true: taken, the includer files it later).
#pragma once
#include <cstddef>

namespace rp {

constexpr uint32_t kChunk = 1024;          // queue arena chunk, entries
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kWgStats = 32;
constexpr uint32_t kArriveSubs = 32;
constexpr uint32_t kPushShards = 8;
constexpr uint32_t kShards = 8;         // of the changed / born record lists (Ctl::chg_n, born_n)
constexpr uint32_t kTgtShards = 8;      // most counters PH_PLACE_BASE hands target ids out of (Cfg::tgt_shards)
constexpr uint32_t kSkip = 0xFFFFFFFEu;     // rec_tgts: the neighbour exists but the record's offer cannot change it (as the record stands)
constexpr int kMaxBuckets = 255;
#ifndef RP_EVQ
#define RP_EVQ 8
#endif
constexpr uint32_t kEvQ = RP_EVQ;          // events of a target one lane of the wave-per-target fold holds
constexpr uint32_t kEvMax = 64 * kEvQ;     // most events a target can hold (Cfg::ev <= this: the capacity in use, and the stride of tgt_ev)
constexpr uint32_t kOwn = 31;              // event code of a record's own pop (LUT indices are 0..25)
constexpr unsigned long long kNever = ~0ull;
constexpr uint32_t kRankBits = 24;
constexpr uint32_t kRankMask = (1u << kRankBits) - 1;

// EsdfVoxel flags in the state word (bits 0-3), parent int8 x/y/z in bits 8-31 — the layout of EsdfDev::state
constexpr uint32_t kObserved = 1, kHallucinated = 2, kInQueue = 4, kFixed = 8;

enum Phase : uint32_t {
  PH_DONE = 0,
  PH_BEGIN,        // control only: pick the next super-step
  PH_PLACE_BASE,   // thread per (base record, position 0..26)
  PH_FOLD,         // thread per dirty target
  PH_APPLY,        // thread per (changed or new record, position)
  PH_SIM,          // thread per excursion whose membership changed
  PH_MINCUT,       // thread per change point
  PH_COMMIT_FOLD,  // thread per target
  PH_RANK,         // SCAN over base records: committed records per base record -> dense pop ranks
  PH_RANK_WRITE,   // thread per record
  PH_PUSH,         // SCAN over committed records in pop order, up to four buckets at a time: FIFO positions, entries written
  PH_CLEANUP,      // thread per target / record
  PH_RAISE_FOLD,   // processRaiseSet: thread per target, one pass (no guessing: a raise pop reads nothing of its own voxel)
};
// A SCAN phase is collective and lives outside this file: for i = 0 .. n_threads - 1 in order, c = rp_scan_count(i),
// rp_scan_apply(i, running sums), running sums += c; the totals go to Ctl::scan_tot.
#ifndef RP_SCANC
#define RP_SCANC 16
#endif
constexpr int kScanC = RP_SCANC;   // components a SCAN phase carries per item: bucket FIFOs filled by one PH_PUSH pass (four until round 5)
struct Cnt4 { uint32_t v[kScanC]; };

struct Cfg {
  float max_distance, min_diff, voxel_size, default_distance;
  int full, multi_queue, num_buckets;
  uint32_t kmax, smax, max_iters;
  uint32_t cut_mult, ramp_mult;   // a bucket's next super-step after a cut takes cut_mult x what got through (at least 32), after a clean one ramp_mult x as many
  uint32_t fold_pairs;   // (device) fold launches over more targets than waves take lists of up to 32 events two at a time, a half-wave each (VBX_RP_FOLD_PAIRS; off in bulk updates)
  uint32_t ev;       // events a target can hold (<= kEvMax); a record whose event does not fit is poisoned and the super-step ends in front of it
  uint32_t filter;   // rp_offer_possible: 0 every offer is an event, 1 + usable neighbours only, 2 + pops that offer at all, 3 + offers that can beat the neighbour
  uint32_t lds_counts;   // (device) 1: COMMIT_FOLD / RAISE_FOLD count pushes and relaxations per workgroup in LDS, one atomic per queue and workgroup
  uint32_t member_limit; // (serial ranking) 1: stop behind a record one of whose children is not in the member list, as the device's ranking must
  uint32_t slot_by_base; // 1: the list of base record i's excursion is slot i + 1 (sub_slots_cap >= kmax); 0: slots are handed out as excursions appear
  uint32_t tgt_claim;    // 1: rp_target claims the voxel before it takes an id (no holes); 0: id first, a lost race leaves a hole
  uint32_t fold_all;     // 1: PH_PLACE_BASE leaves the dirty list alone, the first FOLD of a super-step takes every target; 0: round 5a's lists
  uint32_t stats;        // 1: VBX_RP_STATS — steps per phase are counted and timed, the rankings and folds keep what they cost per workgroup (Args::wg_stats)
  uint32_t tgt_shards;   // target ids in PH_PLACE_BASE come from this many counters, interleaved (id = k * shards + shard; 1: one counter)
  uint32_t mark_moved;   // a ranking marks the targets of 2: the records whose order it changed, 1: every record whose pop time it moved (rp_mark_rec_targets); 0: nothing (round 4)
};

// control block (device memory, one instance).  Part A is written by rp_control only (every workgroup reads it at
// the start of a launch); part B is updated with atomics while a phase runs, and rp_control reads it through RP_LD
// (an atomic read-modify-write: the per-XCD L2s are not coherent, a plain load may see an old line).
struct alignas(128) CtlLine { uint32_t v; uint32_t pad[31]; };
struct Ctl {
  // ---- A
  uint32_t phase, done;
  uint32_t n_threads;                        // threads (items) the current phase needs
  uint32_t bucket, K, base_head;             // super-step: bucket b, base records, FIFO index of base record 0
  uint32_t raise;                            // 1: the super-step pops raise_ (queue num_buckets) instead of a bucket of open_
  uint32_t n_rec, iter;
  uint32_t read;                             // dirty target lists: FOLD reads list `read`, marks go to 1 - read
  uint32_t fold_all;                         // 1: the FOLD phase that follows PH_PLACE_BASE folds targets 0 .. n_threads - 1 (no list: PLACE_BASE does not mark)
  uint32_t a_chg, a_born, a_tgt;             // copies of n_chg / n_born / n_tgt as the last phase left them
  uint32_t chg_pre[kShards + 1], born_pre[kShards + 1];   // where shard s of the changed / born lists starts in PH_APPLY's item numbers (a_chg / a_born: all shards)
  uint32_t cap_stop;                         // 1: the records (or their list) are full — no births any more, the super-step commits what stands
  unsigned long long cut;
  uint32_t n_commit;                         // committed records
  uint32_t scan_tot[kScanC];
  uint32_t push_b[kScanC], push_n;                // buckets of the current PH_PUSH pass
  uint32_t push_next;                        // first bucket not yet handled by a PH_PUSH pass
  uint32_t push_last, scan_n, ord_identity;  // 1: this PH_PUSH pass is the last and cleans up behind itself (no PH_CLEANUP launch); items of the pass's scan; 1: ord[r] = r (no dense ranking ran)
  uint32_t chunk_top;
  // (device wrapper) n_threads | phase << 32 | launch number mod 64 << 40 of the step the NEXT launch runs, stored and read
  // as one word: a workgroup the dispatcher starts after its own launch's control step already ran must not take the next
  // step's phase for its own.  (With part A: every workgroup of a launch reads it, nothing writes here while a phase runs.)
  unsigned long long hdr;
  unsigned long long st_iters;               // iterations so far (PH_APPLY stamps the records it makes with it: Args::rec_born_it)
  unsigned long long t_prev;                 // (device wrapper) clock at the last control step
  // ---- B.  A 128-byte line takes ~87 returning atomics per microsecond NO MATTER HOW MANY WORDS OF IT they address, and lines
  // take them side by side (tools/microbench/lat_bench.hip: 16 counters one word apart 85 tickets per us, 128 bytes apart 850).
  // Until round 6 every counter below sat in the same two lines — the fold's changed / born records, the apply's lists and the
  // workgroups' arrivals queued up behind each other.  One line per counter that a phase hammers.
  alignas(128) uint32_t error;               // 1 record capacity, 2 target capacity, 4 queue arena, 8 no progress, 16 event overflow at base record 0
  uint32_t k_limit;                          // base records from here on cannot take part (event list overflow)
  unsigned long long first_change, smax_cut;
  alignas(128) uint32_t n_tgt;
  alignas(128) uint32_t n_dirty[2];
  alignas(128) uint32_t n_sd;
  alignas(128) uint32_t n_cp;
  alignas(128) uint32_t arrive;              // (device wrapper) workgroups that finished the phase
  // ---- per queue (bucket 0 .. num_buckets - 1, raise_ = num_buckets); the device wrapper moves the first num_buckets + 1 of each
  uint32_t head[kMaxBuckets + 1], tail[kMaxBuckets + 1];  // A: FIFO indices (entries ever popped / pushed)
  uint32_t reserved[kMaxBuckets + 1];        // A: chunks of the queue's FIFO that are backed by the arena
  uint32_t k_cur[kMaxBuckets + 1];           // A: base records the bucket's next super-step may take (slow start after a cut)
  uint32_t push_cnt[kMaxBuckets + 1];        // B: pushes per queue seen by COMMIT_FOLD (upper bound of what gets queued)
  // (device wrapper) arrivals in two levels: workgroup w arrives at counter w % kArriveSubs, the last one there at `arrive` — 1,024
  // arrivals on one line are 12 us (lat_bench's step kernel: 16.9 us per launch against 5.8 with 64 workgroups arriving)
  CtlLine arrive_sub[kArriveSubs];
  // B: PH_PLACE_BASE's target ids, shard s hands out s, s + shards, s + 2 shards ... (a super-step of 16 k base records asks for ~100 k ids
  // in one launch; the control step behind it sets n_tgt to the largest id in use + 1 and the later phases go on from there)
  CtlLine tgt_n[kTgtShards];
  // B: entries of the changed / born record lists per shard (RP_SHARD: the device takes the wave's number).  A FOLD launch over
  // 8 k targets appends a few thousand changed records — on one line that is what the launch lasted.  Shard s of a list is
  // Args::chg + s * rec_cap (born: * 6); PH_APPLY numbers its items through the shards one behind the other (chg_pre / born_pre)
  CtlLine chg_n[kShards], born_n[kShards];
  // ---- statistics, LAST: the device's control step does not load them — it starts from zeros in its LDS copy and ADDS what it
  // counted to these words when it stores the block back (a third of the block's words, and a launch pays for every word the
  // control step moves).  What decides anything is not here: st_iters and t_prev are in part A.
  alignas(128) unsigned long long st_raise_pops, st_raise_steps, st_pops, st_relax, st_supersteps, st_folds, st_exc, st_cut_iters, st_cut_smax, st_steps, st_poison, st_trunc_q, st_trunc_rank, st_retries;
  unsigned long long st_phase_steps[16], st_phase_threads[16], st_phase_ticks[16];
  unsigned long long st_ctl_ticks[4];   // (device) the control step by stage: arrival -> copy in, rp_control, copy out (10 ns units)
  unsigned long long st_bin_steps[8][8], st_bin_ticks[8][8];   // (device) launches of FOLD / APPLY / SIM / PLACE / PUSH / COMMIT_FOLD / CLEANUP / RAISE_FOLD by items: < 4, < 16, < 64, < 256, < 1 Ki, < 4 Ki, < 16 Ki, more
};

static_assert(offsetof(Ctl, error) % 128 == 0 && offsetof(Ctl, n_tgt) % 128 == 0 && offsetof(Ctl, n_dirty) % 128 == 0 && offsetof(Ctl, chg_n) % 128 == 0 &&
              offsetof(Ctl, born_n) % 128 == 0 && offsetof(Ctl, n_sd) % 128 == 0 && offsetof(Ctl, n_cp) % 128 == 0 && offsetof(Ctl, arrive) % 128 == 0 &&
              offsetof(Ctl, arrive_sub) % 128 == 0 && offsetof(Ctl, tgt_n) % 128 == 0 && offsetof(Ctl, st_raise_pops) % 128 == 0,
              "a counter that a phase hammers has a 128-byte line of its own");
static_assert(offsetof(Ctl, hdr) < offsetof(Ctl, error), "the header is read by every workgroup of a launch: with part A, not on a line that takes atomics");

struct Args {
  Cfg c;
  Ctl* ctl;
  // the ESDF layer (pool slot * nvox + linear index) and block adjacency
  float* dist;
  uint32_t* state;
  const uint32_t* nbslot;   // [slot][27]: pool slot of the ESDF block at offset (dx,dy,dz) = ((k%3)-1, (k/3%3)-1, (k/9)-1), kNone if the ESDF layer has none
  const uint8_t* hazard;    // per voxel: a neighbour of the other sign class exists (null: no offer is ever left out)
  uint32_t* blk_dirty;      // per slot word that gets `dirty_bit` or-ed in when a voxel of the block changes (may be null)
  uint32_t dirty_bit;
  uint32_t nvox;
  int vps;
  // bucket FIFOs
  uint32_t* arena;
  uint32_t* chunk_tab;      // [bucket][max_chunks]
  uint32_t max_chunks;
  // records
  uint32_t rec_cap;
  uint32_t* rec_vox;
  uint32_t* rec_pusher;     // kNone: base record
  uint32_t* rec_base;       // the base record whose excursion the record belongs to (itself for a base record)
  uint32_t* rec_meta;       // lut | bucket << 8 | live << 16; "next" copy in rec_meta_n (bucket, live)
  uint32_t* rec_meta_n;
  uint32_t* rec_poison;     // 1: a target could not take the record's event — the record stays out of every super-step fold
  unsigned long long* rec_T;
  float* rec_d;             // voxel distance at pop time (current guess) / next
  float* rec_d_n;
  uint32_t* rec_s;          // voxel state word at pop time
  uint32_t* rec_s_n;
  uint32_t* rec_kid;        // [rec][26]: excursion record + 1 spawned by this record's push to LUT neighbour k
  uint32_t* rec_tgts;       // [rec][27]: targets of the 26 neighbours and (26) of the voxel itself
  uint32_t* rec_push;       // [rec][26] bytes packed in 7 words: bucket + 1 of the committed push to neighbour k
  uint32_t* rec_born_it;    // (may be null) low word of Ctl::st_iters when PH_APPLY made the record: the ranking of that very iteration need not
                            // mark its targets (rp_place just did)
  // targets
  uint32_t tgt_cap;
  uint32_t* vox2tgt;        // [pool voxels]: target + 1
  uint32_t* tgt_gid;
  uint32_t* tgt_cnt;
  uint32_t* tgt_ev;         // [tgt][Cfg::ev]: record << 5 | code
  uint32_t* tgt_dirty;
  uint32_t* dl[2];          // dirty target lists
  // per-iteration lists
  uint32_t* chg;            // records whose pop-time state / liveness changed
  uint32_t* born;           // [.][6]: pusher, lut, bucket, gid, first guess of the pop-time distance / state
  uint32_t* cp;             // change points (records)
  uint32_t* sd_list;        // base records whose excursion must be re-ranked
  uint32_t* sub_dirty;      // [kmax]
  uint32_t* sub_n;          // [kmax] ranked excursion records of the base record
  uint32_t* sub_slot;       // [kmax] slot + 1 of the base record's list in sub_list
  uint32_t* sub_list;       // [slots][smax] ranked records of the last ranking
  uint32_t* sub_slots_used;
  uint32_t* sim_old;        // (serial form, may be null) [slots][smax] the list of the last ranking while the next one is made (Cfg::mark_moved = 2)
  // (device only, may be null) members of every excursion in birth order, for the wave-cooperative ranking
  uint32_t* sub_mem;        // [slots][smax]
  uint32_t* sub_mem_n;      // [kmax]
  uint32_t* rec_local;      // [rec] 1 + index in its excursion's member list (0: the base record)
  uint32_t* push_shards;    // (device, Cfg::lds_counts) [kPushShards][kMaxBuckets + 2]: the workgroups' push counts per queue and, last, their relaxations, by workgroup number —
                            // 1,024 workgroups x a handful of queues on ONE line of Ctl::push_cnt were what a large COMMIT_FOLD launch lasted; the control step sums the shards into Ctl::push_cnt and zeroes them
  unsigned long long* wg_stats;   // (VBX_RP_STATS, may be null) [workgroup][kWgStats]: what a ranking cost, kept per workgroup — atomics on Ctl's lines slow down what they measure
  uint32_t* rec_plocal;     // [rec] rec_local of the record's pusher, noted at birth (may be null: the ranking then asks rec_local[rec_pusher[r]] — one more dependent trip to memory per member)
  uint32_t* sub_restart;    // [kmax] smallest rank at which the excursion's structure changed since its last ranking
  unsigned long long* sim_q;  // [slots][smax] scratch of the ranking
  uint32_t sub_slots_cap;
  // commit
  uint32_t* ord;            // committed records in pop order
  uint32_t* off0;           // [kmax] dense rank of base record i
};

// ------------------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------------------
RP_FN int rp_signum(float v) { return (v > 0.f) - (v < 0.f); }
RP_FN uint32_t rp_pack_parent(int x, int y, int z) {
  return ((uint32_t)(x & 0xFF) << 8) | ((uint32_t)(y & 0xFF) << 16) | ((uint32_t)(z & 0xFF) << 24);
}
RP_FN void rp_unpack_parent(uint32_t s, int* x, int* y, int* z) {
  *x = (int)(int8_t)(s >> 8); *y = (int)(int8_t)(s >> 16); *z = (int)(int8_t)(s >> 24);
}
RP_FN void rp_lut_offset(int idx, int* dx, int* dy, int* dz) {  // neighbor_tools.cc:8-34, column order
  // 6 faces, 12 edges, 8 corners: per axis the 26 offsets + 1 as 2-bit fields of one 64-bit literal (entry idx at bits
  // 2 idx .. 2 idx + 1).  Pure ALU: the byte table this replaces was a global load followed by a wait in every iteration of the
  // folds' event loops (one dependent trip to memory per event of a target).
  //   idx:  0 1 2 3 4 5 | 6 7 8 9 10 11 12 13 14 15 16 17 | 18 .. 25
  //   x+1:  0 2 1 1 1 1 | 0 0 2 2 1  1  1  1  0  2  0  2  | 0 0 0 0 2 2 2 2
  //   y+1:  1 1 0 2 1 1 | 0 2 0 2 0  0  2  2  1  1  1  1  | 0 0 2 2 0 0 2 2
  //   z+1:  1 1 1 1 0 2 | 1 1 1 1 0  2  0  2  0  0  2  2  | 0 2 0 2 0 2 0 2
  const unsigned long long X = 0xaa008855a0558ull, Y = 0xa0a055a088585ull, Z = 0x8888a08855855ull;
  const unsigned sh = 2u * (unsigned)idx;
  *dx = (int)((X >> sh) & 3ull) - 1; *dy = (int)((Y >> sh) & 3ull) - 1; *dz = (int)((Z >> sh) & 3ull) - 1;
}
RP_FN float rp_lut_distance(int idx) {
  const float sq2 = (float)1.4142135623730951, sq3 = (float)1.7320508075688772;  // std::sqrt(2), std::sqrt(3) as float
  return idx < 6 ? 1.0f : (idx < 18 ? sq2 : sq3);
}
RP_FN float rp_parent_norm(int x, int y, int z) {  // Eigen Vector3i::cast<float>().norm(): sqrt(c0 + (c1 + c2))
  const float fx = (float)x, fy = (float)y, fz = (float)z;
  return sqrtf(fx * fx + (fy * fy + fz * fz));
}
// BucketQueue::push's bucket (bucket_queue.h:41-50)
RP_FN int rp_bucket_of(const Cfg& c, float value_f) {
  double value = (double)value_f;
  const double max_val = (double)c.max_distance;
  if (value > max_val) value = max_val;
  int b = (int)floor(fabs(value) / max_val * (double)(c.num_buckets - 1));
  if (b >= c.num_buckets) b = c.num_buckets - 1;
  if (b < 0) b = 0;
  return b;
}
// the voxel `idx` (LUT) away from gid, kNone when the ESDF layer has no block there
RP_FN uint32_t rp_neighbour(const Args& a, uint32_t gid, int idx) {
  const uint32_t slot = gid / a.nvox, lin = gid % a.nvox;
  const int vps = a.vps;
  int dx, dy, dz;
  rp_lut_offset(idx, &dx, &dy, &dz);
  int nx = (int)(lin % vps) + dx, ny = (int)((lin / vps) % vps) + dy, nz = (int)(lin / (vps * vps)) + dz;
  int cx = 1, cy = 1, cz = 1;
  if (nx < 0) { nx += vps; cx = 0; } else if (nx >= vps) { nx -= vps; cx = 2; }
  if (ny < 0) { ny += vps; cy = 0; } else if (ny >= vps) { ny -= vps; cy = 2; }
  if (nz < 0) { nz += vps; cz = 0; } else if (nz >= vps) { nz -= vps; cz = 2; }
  uint32_t s2 = slot;
  if (cx != 1 || cy != 1 || cz != 1) {
    s2 = a.nbslot[(size_t)slot * 27 + (cx + 3 * cy + 9 * cz)];
    if (s2 == kNone) return kNone;
  }
  return s2 * a.nvox + (uint32_t)(nx + vps * (ny + vps * nz));
}
RP_FN uint32_t rp_queue_entry(const Args& a, int q, uint32_t idx) {
  return a.arena[(size_t)a.chunk_tab[(size_t)q * a.max_chunks + (idx / kChunk)] * kChunk + (idx % kChunk)];
}
RP_FN void rp_queue_store(const Args& a, int q, uint32_t idx, uint32_t v) {
  a.arena[(size_t)a.chunk_tab[(size_t)q * a.max_chunks + (idx / kChunk)] * kChunk + (idx % kChunk)] = v;
}
// chunks for FIFO indices [tail, tail + n) of queue q (control thread only)
RP_FN bool rp_queue_reserve(const Args& a, int q, uint32_t n) {
  Ctl& c = *a.ctl;
  if (n == 0) return true;
  const uint32_t last = (c.tail[q] + n - 1) / kChunk;
  for (uint32_t j = c.reserved[q]; j <= last; ++j) {
    if (c.chunk_top >= a.max_chunks || j >= a.max_chunks) { c.error |= 4; return false; }
    a.chunk_tab[(size_t)q * a.max_chunks + j] = c.chunk_top++;
  }
  if (last + 1 > c.reserved[q]) c.reserved[q] = last + 1;
  return true;
}

RP_FN uint32_t rp_meta_lut(uint32_t m) { return m & 0xFF; }
RP_FN uint32_t rp_meta_bucket(uint32_t m) { return (m >> 8) & 0xFF; }
RP_FN bool rp_meta_live(uint32_t m) { return (m >> 16) & 1; }
RP_FN uint32_t rp_meta(uint32_t lut, uint32_t bucket, bool live) { return lut | (bucket << 8) | ((uint32_t)live << 16); }

// item i of a sharded list -> its place (pre[]: where the shards start in item numbers; RP_PRE reads it — the device keeps a
// copy in LDS for the phase, three reads decide among eight shards)
#ifndef RP_PRE
#define RP_PRE(arr, k) (arr)[k]
#endif
RP_FN size_t rp_shard_pos(const uint32_t* pre, uint32_t i, uint32_t cap) {
  static_assert(kShards == 8, "three halvings");
  uint32_t sh = i >= RP_PRE(pre, 4) ? 4u : 0u;
  sh += i >= RP_PRE(pre, sh + 2) ? 2u : 0u;
  sh += i >= RP_PRE(pre, sh + 1) ? 1u : 0u;
  return (size_t)sh * cap + (i - RP_PRE(pre, sh));
}
// appends to the changed / born record lists, on the caller's shard
RP_FN void rp_chg_push(const Args& a, uint32_t r) {
  const uint32_t sh = (uint32_t)(RP_SHARD) % kShards;
  const uint32_t k = atomicAdd(&a.ctl->chg_n[sh].v, 1u);
  if (k < a.rec_cap) a.chg[(size_t)sh * a.rec_cap + k] = r;   // (a record is listed once or twice per iteration)
  else atomicOr(&a.ctl->error, 1u);
}
// a place in the born list for a push of record `pusher` (null: no room even to note it — the cut falls in front of the pusher)
RP_FN uint32_t* rp_born_slot(const Args& a, uint32_t pusher) {
  const uint32_t sh = (uint32_t)(RP_SHARD) % kShards;
  const uint32_t k = atomicAdd(&a.ctl->born_n[sh].v, 1u);
  if (k >= a.rec_cap) { atomicMin(&a.ctl->first_change, a.rec_T[pusher]); return nullptr; }
  return a.born + ((size_t)sh * a.rec_cap + k) * 6;
}
// the block of voxel gid has changed.  A look first: the bit is set once per block and update, and a fold per committed voxel
// or-ing it in again is a few hundred thousand atomics per update on the half dozen lines that hold the blocks' words (a look
// that misses the bit — another XCD's L2 may hold the old line — sets it once more: harmless)
RP_FN void rp_mark_block(const Args& a, uint32_t gid) {
  if (!a.blk_dirty) return;
  uint32_t* w = &a.blk_dirty[gid / a.nvox];
  if ((*w & a.dirty_bit) != a.dirty_bit) atomicOr(w, a.dirty_bit);
}
RP_FN void rp_mark_dirty(const Args& a, uint32_t t) {
  if (t >= kSkip) return;
  Ctl& c = *a.ctl;
  // every target PH_PLACE_BASE touches is new and will be folded: no flag, no list entry (a flag and a slot of one shared
  // counter per placement — tens of thousands of atomics on one cache line in a super-step of 16 k base records)
  if (c.phase == PH_PLACE_BASE && a.c.fold_all) return;
  if (atomicExch(&a.tgt_dirty[t], 1u) == 0u) {
    // (device) the workgroup collects its marks and takes ONE range of the list for all of them when the phase ends
    // (k_rp_step): Ctl::n_dirty is a single word that every wave with a mark used to increment — a hundred rankings per
    // launch, eight increments each, on an address that takes ~90 atomics per microsecond: two thirds of a ranking's time
#ifdef RP_WG_DIRTY_PUSH
    if (RP_WG_DIRTY_PUSH(t)) return;
#endif
    const uint32_t w = 1u - c.read;
    const uint32_t k = RP_INC(&c.n_dirty[w]);
    a.dl[w][k] = t;   // k < tgt_cap: a target is listed at most once per list
  }
}

// the target of voxel gid (created on first use)
//
// Cfg::tgt_claim: the voxel is CLAIMED first (vox2tgt 0 -> kBusyTgt), only the claimant takes an id, writes it over the claim,
// and whoever finds the claim looks again until the id is there.  The records of a super-step are neighbours of each other:
// up to 27 lanes want the same new target at the same moment, and when each of them took an id before trying to publish it
// (rounds 4 / 5a, tgt_claim = 0) more than half of all ids ended as holes — 1.14 M ids for 0.5 M targets per update of the
// configs[3] stream — which every thread-per-target phase (first fold, commit fold, raise fold, clean-up) then walks.  Inside a
// wave the claimant's branch runs to its end before the loop goes round again, so nobody waits for a lane that cannot move;
// a wait that does not end (kTgtSpin looks) gives the target up like a full pool does (the record is poisoned).
constexpr uint32_t kBusyTgt = 0xFFFFFFFFu;
constexpr uint32_t kTgtSpin = 1u << 14;
// (v: what vox2tgt[gid] held when the caller looked — rp_place reads it together with the voxel's state, distance and hazard
// byte, one trip to memory instead of four in a row)
#ifndef RP_LD_RO
#define RP_LD_RO(x) RP_LD(x)
#endif
// a new target id.  PH_PLACE_BASE: from the caller's shard of Cfg::tgt_shards interleaved sequences (the ids a shard leaves unused
// below the largest one stay holes: tgt_gid is kNone there — PH_CLEANUP leaves it so)
RP_FN uint32_t rp_target_id(const Args& a) {
  Ctl& c = *a.ctl;
  const uint32_t S = a.c.tgt_shards;
  if (S > 1u && c.phase == PH_PLACE_BASE) {
    const uint32_t sh = (uint32_t)(RP_SHARD) % S;
    return RP_INC(&c.tgt_n[sh].v) * S + sh;
  }
  return RP_INC(&c.n_tgt);
}
RP_FN uint32_t rp_target(const Args& a, uint32_t gid, uint32_t v) {
  if (v != 0u && v != kBusyTgt) return v - 1u;
  if (a.c.tgt_claim) {
    for (uint32_t spin = 0; spin < kTgtSpin; ++spin) {
      // (a voxel somebody else is busy with is WATCHED with reads — RP_LD_RO: a coherent load where the includer has one —
      // and only asked for with a compare-and-swap again once it reads free: up to 26 lanes wait for the 27th here, and a
      // swap per look is a read-modify-write at the memory side where a look costs a read)
      if (v == kBusyTgt) {
        v = RP_LD_RO(a.vox2tgt[gid]);
        if (v == kBusyTgt) continue;
        if (v != 0u) return v - 1u;
      }
      const uint32_t old = atomicCAS(&a.vox2tgt[gid], 0u, kBusyTgt);
      v = old;
      if (old == 0u) {
        const uint32_t id = rp_target_id(a);
        if (id >= a.tgt_cap) {
          atomicExch(&a.vox2tgt[gid], 0u);
          return kNone;
        }
        a.tgt_gid[id] = gid;
        atomicExch(&a.vox2tgt[gid], id + 1u);
        return id;
      }
      if (old != kBusyTgt) return old - 1u;
    }
    return kNone;
  }
  const uint32_t id = rp_target_id(a);
  if (id >= a.tgt_cap) return kNone;
  const uint32_t old = atomicCAS(&a.vox2tgt[gid], 0u, id + 1u);
  if (old != 0u) {  // somebody else made it: `id` stays an empty hole
    a.tgt_gid[id] = kNone;
    return old - 1u;
  }
  a.tgt_gid[id] = gid;
  return id;
}

// Can the offer of a record whose voxel stands at (vd, vs) when it pops change neighbour ngid at all?  Leaving out
// offers that cannot keeps two thirds of the targets from existing.  A neighbour that is unobserved or fixed is never
// written (:414-417).  The sign class of a voxel (d > 0 or not) never changes inside processOpenSet (every branch of
// :431-491 writes a distance of the neighbour's own class), so for a neighbour of the record's class with no neighbour
// of the other class anywhere around it (Args::hazard) only the two same-sign branches can ever fire, they only move
// the neighbour towards zero, and an offer that does not beat the neighbour's distance at the start of the super-step
// cannot beat a later one.  The test is repeated whenever the record's guessed state changes (rp_phase_apply).
// (sn, nd, hz: the neighbour's state word, distance and hazard byte, read by the caller in one go)
RP_FN bool rp_offer_possible(const Args& a, float vd, uint32_t vs, uint32_t sn, float nd, bool hz, int lut) {
  if (a.c.filter < 1) return true;
  if (!(sn & kObserved) || (sn & kFixed)) return false;
  if (a.ctl->raise || a.c.full || !a.hazard || a.c.filter < 2) return true;
  if (!(vs & kObserved) || vd >= a.c.max_distance || vd <= -a.c.max_distance) return false;   // the pop offers nothing (:389-392)
  if (a.c.filter < 3) return true;
  if ((vd > 0) != (nd > 0)) return true;
  if (hz) return true;
  const float distance = rp_lut_distance(lut) * a.c.voxel_size;
  return vd > 0 ? (vd + distance + a.c.min_diff < nd) : (vd - distance - a.c.min_diff > nd);
}

// record `r` (voxel gid, guessed pop-time state vd / vs) announces itself to the target at position p (0..25 LUT
// neighbour, 26 the voxel itself)
RP_FN void rp_place(const Args& a, uint32_t r, uint32_t base_rec, uint32_t gid, uint32_t p, float vd, uint32_t vs) {
  Ctl& c = *a.ctl;
  const uint32_t ngid = p == 26 ? gid : rp_neighbour(a, gid, (int)p);
  uint32_t t = kNone;
  if (ngid != kNone) {
    // everything this thread will want to know about the voxel, asked for at once (the loads are independent; behind the
    // early returns of the tests below they were four dependent trips)
    const uint32_t v2t = a.vox2tgt[ngid];
    const uint32_t sn = a.state[ngid];
    const float nd = a.dist[ngid];
    const bool hz = a.hazard ? a.hazard[ngid] != 0 : true;
    if (p != 26 && !rp_offer_possible(a, vd, vs, sn, nd, hz, (int)p)) {
      a.rec_tgts[(size_t)r * 27 + p] = kSkip;
      return;
    }
    t = rp_target(a, ngid, v2t);
  }
  a.rec_tgts[(size_t)r * 27 + p] = t;
  if (ngid == kNone) return;
  const uint32_t k = t == kNone ? a.c.ev : atomicAdd(&a.tgt_cnt[t], 1u);   // (no target to be had: as if its list were full)
  if (k < a.c.ev) {
    a.tgt_ev[(size_t)t * a.c.ev + k] = (r << 5) | (p == 26 ? kOwn : p);
  } else if (atomicExch(&a.rec_poison[r], 1u) == 0u) {
    // the target cannot hear this record: the record must stay out of the super-step.  A base record stops the
    // super-step in front of itself; an excursion record that has a rank already (its offer is placed late, after its
    // guessed state moved) is ranked again — the ranking ends in front of a poisoned record.
    atomicAdd(&c.st_poison, 1ull);
    if (r < c.K) {
      atomicMin(&c.k_limit, r);
    } else {
      // (base_rec: a record whose excursion is r's and whose fields stand — r itself, or its pusher while PH_APPLY is still making
      // r: a record's own fields may be on their way when a sibling thread gets here.  Looked up here, where it is needed: as an
      // argument it was a trip to memory in front of every placement of a new record)
      const uint32_t base = a.rec_base[base_rec];
      const unsigned long long T = a.rec_T[r];
      if (a.sub_restart && T != kNever && (T & kRankMask) != 0) atomicMin(&a.sub_restart[base], (uint32_t)(T & kRankMask) - 1u);
      if (atomicExch(&a.sub_dirty[base], 1u) == 0u) a.sd_list[atomicAdd(&c.n_sd, 1u)] = base;
    }
  }
  rp_mark_dirty(a, t);
}

// ------------------------------------------------------------------------------------------------------------
// the relaxation of one neighbour, esdf_integrator.cc:405-491.  (vd, vs): the popped voxel at its pop time; (nd, ns):
// the neighbour now.  Returns true and the new distance / parent bits if the neighbour is written.
// ------------------------------------------------------------------------------------------------------------
RP_FN bool rp_relax(const Cfg& c, float vd, uint32_t vs, float nd, int lut, float* out_d, uint32_t* out_parent) {
  int dx, dy, dz;
  rp_lut_offset(lut, &dx, &dy, &dz);
  float distance = rp_lut_distance(lut) * c.voxel_size;
  int npx = -dx, npy = -dy, npz = -dz;
  if (c.full) {  // :419-428
    int vpx, vpy, vpz;
    rp_unpack_parent(vs, &vpx, &vpy, &vpz);
    npx = vpx - dx; npy = vpy - dy; npz = vpz - dz;
    distance = c.voxel_size * (rp_parent_norm(npx, npy, npz) - rp_parent_norm(vpx, vpy, vpz));
    if ((double)distance < 0.0) return false;
  }
  float new_d;
  if (vd > 0 && nd > 0) {                                        // :431-444
    if (!(vd + distance + c.min_diff < nd)) return false;
    new_d = vd + distance;
  } else if (vd <= 0 && nd <= 0) {                               // :446-459
    if (!(vd - distance - c.min_diff > nd)) return false;
    new_d = vd - distance;
  } else {                                                       // :461-491
    const float potential = vd - (float)rp_signum(vd) * distance;
    if (!(fabsf(potential - nd) > distance)) return false;
    if ((float)rp_signum(potential) == nd) new_d = potential;    // :464 compares signum(int) with the float distance
    else new_d = (float)rp_signum(nd) * distance;
  }
  *out_d = new_d;
  *out_parent = rp_pack_parent(npx, npy, npz);
  return true;
}

// ------------------------------------------------------------------------------------------------------------
// the fold of one target.  limit: only events with T < limit.  commit: write the voxel and the records' pushes
// instead of the iteration's outputs.
// ------------------------------------------------------------------------------------------------------------
struct FoldEv {
  unsigned long long T;
  uint32_t code;  // record << 5 | lut
};

constexpr uint32_t kLp = 16;     // pushes below b one fold can see (one per pop of the voxel, plus one)

RP_FN void rp_fold(const Args& a, uint32_t t, unsigned long long limit, bool commit) {
  Ctl& c = *a.ctl;
  const uint32_t gid = a.tgt_gid[t];
  if (gid == kNone) return;
  const int b = (int)c.bucket;
  uint32_t n_all = a.tgt_cnt[t];
  if (n_all > a.c.ev) n_all = a.c.ev;
  FoldEv ev[kEvMax];
  uint32_t n = 0;
  for (uint32_t k = 0; k < n_all; ++k) {
    const uint32_t code = a.tgt_ev[(size_t)t * a.c.ev + k];
    const uint32_t r = code >> 5;
    const uint32_t m = a.rec_meta[r];
    const unsigned long long T = a.rec_T[r];
    if (!rp_meta_live(m) || a.rec_poison[r] || !(T < limit)) continue;
    // insertion by T (events of one record on one target are unique, so are the T)
    uint32_t j = n++;
    while (j > 0 && ev[j - 1].T > T) { ev[j] = ev[j - 1]; --j; }
    ev[j].T = T; ev[j].code = code;
  }
  const float d0 = a.dist[gid];
  const uint32_t s0 = a.state[gid];
  float d = d0;
  uint32_t s = s0;
  const bool usable = (s0 & kObserved) && !(s0 & kFixed);
  uint32_t relax = 0;
  // pushes below b seen in this fold: record, lut | bucket << 8, and the voxel as the push left it
  uint32_t lp_rec[kLp], lp_lb[kLp], lp_s[kLp];
  float lp_d[kLp];
  uint32_t n_lp = 0;
  for (uint32_t k = 0; k < n; ++k) {
    const uint32_t code = ev[k].code;
    const uint32_t r = code >> 5, lut = code & 31;
    if (lut == kOwn) {
      // the pop: processOpenSet reads the voxel here (:381-392)
      if (!commit) {
        a.rec_d_n[r] = d;
        a.rec_s_n[r] = s;
      }
      s &= ~kInQueue;                                            // :386
      continue;
    }
    const float vd = a.rec_d[r];
    const uint32_t vs = a.rec_s[r];
    if (!(vs & kObserved) || vd >= a.c.max_distance || vd <= -a.c.max_distance) continue;  // :389-392
    if (!usable) continue;                                       // :414-417
    float nd;
    uint32_t np;
    if (!rp_relax(a.c, vd, vs, d, (int)lut, &nd, &np)) continue;
    ++relax;
    d = nd;
    s = (s & 0xFFu) | np;
    if (a.c.multi_queue || !(s & kInQueue)) {
      s |= kInQueue;
      const int nb = rp_bucket_of(a.c, nd);
      if (commit) {
        // rec_push[r][lut] = bucket + 1 (bytes; a record's 26 bytes are written by 26 different targets)
        const uint32_t w = r * 7 + lut / 4, sh = (lut % 4) * 8;
        atomicOr(&a.rec_push[w], (uint32_t)(nb + 1) << sh);
        atomicAdd(&c.push_cnt[nb], 1u);
      } else if (nb < b) {
        if (n_lp == kLp) {
          // no room to describe this push: the pushing record leaves the super-step (the cut falls in front of it)
          if (atomicExch(&a.rec_poison[r], 1u) == 0u) {
            atomicAdd(&c.st_poison, 1ull);
            if (r < c.K) atomicMin(&c.k_limit, r);
            const uint32_t base = a.rec_base[r];
            if (atomicExch(&a.sub_dirty[base], 1u) == 0u) a.sd_list[atomicAdd(&c.n_sd, 1u)] = base;
            rp_chg_push(a, r);
          }
          continue;
        }
        lp_rec[n_lp] = r; lp_lb[n_lp] = lut | ((uint32_t)nb << 8);
        lp_d[n_lp] = d; lp_s[n_lp] = s;
        ++n_lp;
      }
    }
  }
  if (commit) {
    if (d != d0 || s != s0) {
      a.dist[gid] = d;
      a.state[gid] = s;
      rp_mark_block(a, gid);
    }
    if (relax) atomicAdd(&c.st_relax, (unsigned long long)relax);
    return;
  }
  // ---- outputs of an iteration
  for (uint32_t k = 0; k < n_all; ++k) {
    const uint32_t code = a.tgt_ev[(size_t)t * a.c.ev + k];
    if ((code & 31) != kOwn) continue;
    const uint32_t r = code >> 5;
    const uint32_t m = a.rec_meta[r];
    if (a.rec_poison[r]) continue;
    bool changed = false;
    uint32_t mn = m;
    if (a.rec_pusher[r] != kNone) {
      // an excursion record lives iff this fold pushed it below b
      bool found = false;
      uint32_t bucket = rp_meta_bucket(m);
      for (uint32_t j = 0; j < n_lp; ++j)
        if (lp_rec[j] == a.rec_pusher[r] && (lp_lb[j] & 0xFF) == rp_meta_lut(m)) { found = true; bucket = lp_lb[j] >> 8; lp_rec[j] = kNone; }
      mn = rp_meta(rp_meta_lut(m), bucket, found) | (m & (1u << 18));
      if (mn != m) changed = true;
    }
    if (rp_meta_live(m) && a.rec_T[r] != kNever) {
      // it popped in this fold: did its pop-time state move?
      if (__float_as_uint(a.rec_d_n[r]) != __float_as_uint(a.rec_d[r]) || a.rec_s_n[r] != a.rec_s[r]) changed = true;
    } else {
      a.rec_d_n[r] = a.rec_d[r];
      a.rec_s_n[r] = a.rec_s[r];
    }
    a.rec_meta_n[r] = mn;
    if (changed) rp_chg_push(a, r);
  }
  for (uint32_t j = 0; j < n_lp; ++j) {
    if (lp_rec[j] == kNone) continue;  // matched an existing record
    // a push below b without a record yet (an existing record for (pusher, lut) sits on THIS voxel and was matched above)
    if (a.rec_kid[(size_t)lp_rec[j] * 26 + (lp_lb[j] & 0xFF)] != 0u) continue;  // (dead record that is not on this list cannot happen; be safe)
    uint32_t* bw = rp_born_slot(a, lp_rec[j]);
    if (!bw) continue;
    bw[0] = lp_rec[j];
    bw[1] = lp_lb[j] & 0xFF;
    bw[2] = lp_lb[j] >> 8;
    bw[3] = gid;
    bw[4] = __float_as_uint(lp_d[j]);   // first guess of the record's pop-time state: the voxel as this push left it
    bw[5] = lp_s[j];
  }
}


// ------------------------------------------------------------------------------------------------------------
// processRaiseSet (esdf_integrator.cc:305-369) for one generation of raise_: the target's events are the pops of its
// neighbours in FIFO order; a pop whose direction is the voxel's parent resets and raises it, any other pop queues it in
// open_ if it is not queued.  The popped voxel itself is not read, so one pass is exact.
// ------------------------------------------------------------------------------------------------------------
RP_FN bool rp_raise_event(const Cfg& c, float* d, uint32_t* s, int lut, bool* to_raise) {
  if (!(*s & kObserved) || (*s & kFixed)) return false;          // :333-335
  int dx, dy, dz, px, py, pz;
  rp_lut_offset(lut, &dx, &dy, &dz);
  rp_unpack_parent(*s, &px, &py, &pz);
  bool is_parent = (px == -dx && py == -dy && pz == -dz);
  if (c.full) {  // :340-348: the rounded normalised parent against the direction
    const float n2 = (float)px * (float)px + ((float)py * (float)py + (float)pz * (float)pz);
    float ux = (float)px, uy = (float)py, uz = (float)pz;
    if (n2 > 0.f) {
      const float nn = sqrtf(n2);
      ux = ux / nn; uy = uy / nn; uz = uz / nn;
    }
    is_parent = ((int)roundf(ux) == -dx && (int)roundf(uy) == -dy && (int)roundf(uz) == -dz);
  }
  if (is_parent) {
    *d = (float)rp_signum(*d) * c.default_distance;
    *s &= 0xFFu;  // parent.setZero()
    *to_raise = true;
    return true;
  }
  if (!(*s & kInQueue)) {
    *s |= kInQueue;
    *to_raise = false;
    return true;
  }
  return false;
}

RP_FN void rp_fold_raise(const Args& a, uint32_t t) {
  Ctl& c = *a.ctl;
  const uint32_t gid = a.tgt_gid[t];
  if (gid == kNone) return;
  uint32_t n_all = a.tgt_cnt[t];
  if (n_all > a.c.ev) n_all = a.c.ev;
  FoldEv ev[kEvMax];
  uint32_t n = 0;
  for (uint32_t k = 0; k < n_all; ++k) {
    const uint32_t code = a.tgt_ev[(size_t)t * a.c.ev + k];
    if ((code & 31) == kOwn) continue;
    const unsigned long long T = a.rec_T[code >> 5];
    if (!(T < c.cut)) continue;   // (records behind an event-list overflow wait for the next super-step)
    uint32_t j = n++;
    while (j > 0 && ev[j - 1].T > T) { ev[j] = ev[j - 1]; --j; }
    ev[j].T = T; ev[j].code = code;
  }
  const float d0 = a.dist[gid];
  const uint32_t s0 = a.state[gid];
  float d = d0;
  uint32_t s = s0;
  const int RQ = a.c.num_buckets;
  for (uint32_t k = 0; k < n; ++k) {
    const uint32_t r = ev[k].code >> 5, lut = ev[k].code & 31;
    bool to_raise;
    if (!rp_raise_event(a.c, &d, &s, (int)lut, &to_raise)) continue;
    const int q = to_raise ? RQ : rp_bucket_of(a.c, d);
    const uint32_t w = r * 7 + lut / 4, sh = (lut % 4) * 8;
    atomicOr(&a.rec_push[w], (uint32_t)(q + 1) << sh);
    atomicAdd(&c.push_cnt[q], 1u);
  }
  if (d != d0 || s != s0) {
    a.dist[gid] = d;
    a.state[gid] = s;
    rp_mark_block(a, gid);
  }
}

// ------------------------------------------------------------------------------------------------------------
// phases
// ------------------------------------------------------------------------------------------------------------
RP_FN void rp_phase_place_base(const Args& a, uint32_t tid) {
  Ctl& c = *a.ctl;
  const uint32_t r = tid / 27, p = tid % 27;
  if (r >= c.K) return;
  const uint32_t gid = rp_queue_entry(a, (int)c.bucket, c.base_head + r);
  if (p == 26) {
    a.rec_vox[r] = gid;
    a.rec_pusher[r] = kNone;
    a.rec_base[r] = r;
    a.rec_meta[r] = rp_meta(0, c.bucket, true);
    a.rec_meta_n[r] = a.rec_meta[r];
    a.rec_T[r] = (unsigned long long)r << kRankBits;
    a.rec_d[r] = a.dist[gid];
    a.rec_s[r] = a.state[gid];
    a.rec_d_n[r] = a.rec_d[r];
    a.rec_s_n[r] = a.rec_s[r];
    a.ord[r] = r;   // (raise super-steps commit every record, in FIFO order)
    a.sub_dirty[r] = 0;
    a.sub_n[r] = 0;
    a.sub_slot[r] = 0;
    if (a.sub_mem) { a.sub_mem_n[r] = 0; a.rec_local[r] = 0; a.sub_restart[r] = kNone; }
  }
  rp_place(a, r, r, gid, p, a.dist[gid], a.state[gid]);
}

RP_FN void rp_phase_fold(const Args& a, uint32_t tid) {
  Ctl& c = *a.ctl;
  if (tid >= c.n_threads) return;
  const uint32_t t = c.fold_all ? tid : a.dl[c.read][tid];
  a.tgt_dirty[t] = 0;
  rp_fold(a, t, kNever, false);
}

RP_FN void rp_mark_sub_dirty(const Args& a, uint32_t r) {
  Ctl& c = *a.ctl;
  const uint32_t base = a.rec_base[r];
  if (a.sub_restart) {
    // the excursion's pops up to r's stay as they are (an unranked r decides nothing yet: its children enter when it pops)
    const unsigned long long T = a.rec_T[r];
    if (T != kNever) atomicMin(&a.sub_restart[base], (uint32_t)(T & kRankMask));
  }
  if (atomicExch(&a.sub_dirty[base], 1u) == 0u) a.sd_list[atomicAdd(&c.n_sd, 1u)] = base;
}

RP_FN void rp_phase_apply(const Args& a, uint32_t tid) {
  Ctl& c = *a.ctl;
  const uint32_t item = tid / 27, p = tid % 27;
  if (item < c.a_chg) {
    const uint32_t r = a.chg[rp_shard_pos(c.chg_pre, item, a.rec_cap)];
    if (p == 26) {
      // (bit 18 of rec_meta may be set by a birth in this very phase, after FOLD copied the word to rec_meta_n: it is not
      // part of the comparison — a base record, whose other bits never change, would be sent to its pusher otherwise)
      const uint32_t mn = a.rec_meta_n[r] & ~(1u << 18), m = a.rec_meta[r] & ~(1u << 18);
      uint32_t point = r;
      if (mn != m) {
        // liveness / bucket of an excursion record: the change happened at its pusher's pop
        point = a.rec_pusher[r];
        rp_mark_sub_dirty(a, point);
      }
      a.cp[item] = point;   // (one change point per item: its place is the item's — the counter this list had took one atomic per record of every PH_APPLY, for the sake of the few PH_MINCUTs)
      a.rec_d[r] = a.rec_d_n[r];
      a.rec_s[r] = a.rec_s_n[r];
      if (mn != m) {   // (change the other bits only)
        atomicAnd(&a.rec_meta[r], (1u << 18));
        atomicOr(&a.rec_meta[r], mn);
      }
    }
    const uint32_t t = a.rec_tgts[(size_t)r * 27 + p];
    // (rec_d_n / rec_s_n: the new guess; thread 26 of this record is copying it over the old one right now)
    if (t == kSkip) rp_place(a, r, r, a.rec_vox[r], p, a.rec_d_n[r], a.rec_s_n[r]);   // an offer left out so far may count now
    else rp_mark_dirty(a, t);
    return;
  }
  const uint32_t j = item - c.a_chg;
  if (j >= c.a_born) return;
  const uint32_t r = c.n_rec + j;   // (control checked the capacity)
  const uint32_t* bw = a.born + rp_shard_pos(c.born_pre, j, a.rec_cap) * 6;
  const uint32_t pusher = bw[0], lut = bw[1], bucket = bw[2];
  if (c.cap_stop) {
    // no record for this push: the super-step ends in front of the pop that made it
    if (p == 26) {
      atomicMin(&c.first_change, a.rec_T[pusher]);
      a.cp[item] = pusher;
    }
    return;
  }
  const uint32_t gid = bw[3];
  if (p == 26) {
    a.rec_vox[r] = gid;
    a.rec_pusher[r] = pusher;
    a.rec_base[r] = a.rec_base[pusher];
    a.rec_meta[r] = rp_meta(lut, bucket, true);
    a.rec_meta_n[r] = a.rec_meta[r];
    a.rec_T[r] = kNever;
    a.rec_d[r] = __uint_as_float(bw[4]);
    a.rec_s[r] = bw[5];
    a.rec_d_n[r] = a.rec_d[r];
    a.rec_s_n[r] = a.rec_s[r];
    a.rec_kid[(size_t)pusher * 26 + lut] = r + 1;
    if (a.rec_born_it) a.rec_born_it[r] = (uint32_t)c.st_iters;
    a.cp[item] = pusher;
    rp_mark_sub_dirty(a, pusher);
    if (a.sub_mem) {
      const uint32_t base = a.rec_base[pusher];
      // the excursion's list: slot base + 1 (Cfg::slot_by_base — one list per base record, nothing to allocate), or the next
      // free one (until round 5a: 4,096 slots taken first come first served, and every lane that met an excursion without a
      // slot took one before it knew whether it had won — a super-step of 16 k base records with 3,000 excursions ran out,
      // the rankings behind the last slot stopped at their first record and the super-step was cut)
      uint32_t slot = a.sub_slot[base];
      if (slot == 0u) {
        if (a.c.slot_by_base) {
          slot = base + 1u;
          a.sub_slot[base] = slot;
        } else {
          const uint32_t mine = atomicAdd(a.sub_slots_used, 1u) + 1u;
          const uint32_t old = atomicCAS(&a.sub_slot[base], 0u, mine);
          slot = old ? old : mine;
        }
      }
      if (a.rec_plocal) a.rec_plocal[r] = a.rec_local[pusher];   // (the pusher was born in an earlier iteration: its place in the list stands)
      const uint32_t idx = atomicAdd(&a.sub_mem_n[base], 1u);
      if (slot <= a.sub_slots_cap && idx < a.c.smax) {
        a.sub_mem[(size_t)(slot - 1) * a.c.smax + idx] = r;
        a.rec_local[r] = idx + 1;
      } else {
        // not in the list: the ranking stops where this record's pusher pops
        a.rec_local[r] = 0;
        atomicOr(&a.rec_meta[pusher], 1u << 18);
      }
    }
  }
  rp_place(a, r, pusher, gid, p, __uint_as_float(bw[4]), bw[5]);
}

// A ranking that moves a record's pop time changes the ORDER of the events on the targets that record talks to.  PH_APPLY
// marks the targets of the records whose state or liveness changed; the records behind them in the excursion move with
// them, and when one of those overtakes a record of its own excursion on a target both talk to (a child that changed
// its bucket takes its descendants along), or stops popping because the ranking now ends in front of it (smax, a poisoned
// record), nobody told that target.  Found with the emulation's EOM_CHECK (tools/esdf_order_model.cc: at every fixed point
// ALL targets are folded once more and must not change anything in front of the cut): one super-step of the first frame
// of the configs[3] stream — 1 of 2 187 fixed points in six frames — was not consistent; the layer still came out right
// (the missed event did not change the voxel), with this marking all are consistent.  6 % more folds.
// does a ranking of this iteration have to mark r's targets when it moves r's pop time?  Not those of a record PH_APPLY made in this
// very iteration: every one of them is in the dirty list already
RP_FN bool rp_moved_needs_mark(const Args& a, uint32_t r) {
  return !(a.rec_born_it && a.rec_born_it[r] == (uint32_t)a.ctl->st_iters && a.rec_pusher[r] != kNone);
}
RP_FN void rp_mark_rec_targets(const Args& a, uint32_t r) {
  if (!rp_moved_needs_mark(a, r)) return;
  for (uint32_t p = 0; p < 27; ++p) rp_mark_dirty(a, a.rec_tgts[(size_t)r * 27 + p]);
}

// pop times of the excursion of base record `base`: the reference's queue discipline (lowest bucket first, FIFO
// inside, bucket_queue.h:58-80) over the live excursion records; children enter in LUT order when their pusher pops
RP_FN void rp_phase_sim(const Args& a, uint32_t tid) {
  Ctl& c = *a.ctl;
  if (tid >= c.n_threads) return;
  const uint32_t base = a.sd_list[tid];
  a.sub_dirty[base] = 0;
  uint32_t slot = a.sub_slot[base];
  if (slot == 0) {
    slot = a.c.slot_by_base ? base + 1u : atomicAdd(a.sub_slots_used, 1u) + 1u;
    if (slot > a.sub_slots_cap) { atomicOr(&c.error, 1u); return; }
    a.sub_slot[base] = slot;
  }
  uint32_t* list = a.sub_list + (size_t)(slot - 1) * a.c.smax;
  unsigned long long* q = a.sim_q + (size_t)(slot - 1) * a.c.smax;   // pending entries: record | next entry << 32
  const uint32_t old_n = a.sub_n[base];
  // Cfg::mark_moved = 2: only what changes the ORDER of events on some target is marked — a record that pops now and did not
  // before (unless PH_APPLY made it in this iteration), one that popped and does not any more, and one that is overtaken: a
  // record with a larger old rank pops in front of it now (every pair that swapped has such a member, and the targets the pair
  // shares are among that member's).  Ranks that merely shift because something was inserted or removed in front of them move
  // no order.  The old pop times stay in rec_T until the record pops again; the old list is kept aside (Args::sim_old).
  uint32_t* oldl = (a.c.mark_moved >= 2 && a.sim_old) ? a.sim_old + (size_t)(slot - 1) * a.c.smax : nullptr;
  uint32_t runmax = 0;
  if (oldl) for (uint32_t k = 0; k < old_n; ++k) oldl[k] = list[k];
  else for (uint32_t k = 0; k < old_n; ++k) a.rec_T[list[k]] = kNever;
  // one FIFO per bucket below b, linked through q; lowest = a lower bound of the lowest non-empty bucket
  unsigned short head[kMaxBuckets + 1], tail[kMaxBuckets + 1];
  const int nb = (int)c.bucket;
  for (int k = 0; k < nb; ++k) head[k] = tail[k] = 0xFFFF;
  uint32_t n_q = 0, rank = 0;
  int lowest = nb;
  bool truncated = false;
  uint32_t cur = base;
  for (;;) {
    // (Cfg::member_limit: the device ranks from the member lists PH_APPLY keeps — smax births per excursion, dead records
    // included — and has to stop behind a record that has a child outside the list; the serial form can do the same, which
    // is how the emulation reproduces the device's cuts)
    if (a.c.member_limit && a.sub_mem && (a.rec_meta[cur] & (1u << 18))) { truncated = true; atomicAdd(&c.st_trunc_q, 1ull); break; }
    // the children of `cur` enter their buckets in LUT order
    for (int lut = 0; lut < 26 && !truncated; ++lut) {
      const uint32_t kid = a.rec_kid[(size_t)cur * 26 + lut];
      if (kid == 0u) continue;
      const uint32_t m = a.rec_meta[kid - 1];
      if (!rp_meta_live(m)) continue;
      if (n_q >= a.c.smax || n_q >= 0xFFFFu) { truncated = true; atomicAdd(&c.st_trunc_q, 1ull); break; }
      const int kb = (int)rp_meta_bucket(m);
      q[n_q] = (unsigned long long)(kid - 1) | (0xFFFFull << 32);
      if (tail[kb] == 0xFFFF) head[kb] = (unsigned short)n_q;
      else q[tail[kb]] = (q[tail[kb]] & 0xFFFFFFFFull) | ((unsigned long long)n_q << 32);
      tail[kb] = (unsigned short)n_q;
      if (kb < lowest) lowest = kb;
      ++n_q;
    }
    if (truncated) break;
    while (lowest < nb && head[lowest] == 0xFFFF) ++lowest;   // BucketQueue::front / pop
    if (lowest >= nb) break;
    const unsigned long long e = q[head[lowest]];
    const uint32_t r = (uint32_t)(e & 0xFFFFFFFFull);
    head[lowest] = (unsigned short)(e >> 32);
    if (head[lowest] == 0xFFFF) tail[lowest] = 0xFFFF;
    if (rank >= a.c.smax - 1 || a.rec_poison[r]) { truncated = true; atomicAdd(&c.st_trunc_rank, 1ull); break; }
    ++rank;
    if (oldl) {
      const unsigned long long oT = a.rec_T[r];
      const uint32_t o = oT == kNever ? 0u : (uint32_t)(oT & kRankMask);
      if (o == 0u || o < runmax) rp_mark_rec_targets(a, r);
      if (o > runmax) runmax = o;
    }
    a.rec_T[r] = ((unsigned long long)base << kRankBits) | rank;
    if (!oldl && a.c.mark_moved && (rank - 1 >= old_n || list[rank - 1] != r)) {
      // its pop time moved (or it pops for the first time, or again): the targets it talks to fold again
      rp_mark_rec_targets(a, r);
      if (rank - 1 < old_n) rp_mark_rec_targets(a, list[rank - 1]);   // (the record that had this rank: it moves, or it does not pop any more)
    }
    list[rank - 1] = r;
    cur = r;
  }
  if (oldl) {
    // ranked before and not now: its old pop time is still there and it is not where it was in the new list
    for (uint32_t k = 0; k < old_n; ++k) {
      const uint32_t r = oldl[k];
      if (a.rec_T[r] == (((unsigned long long)base << kRankBits) | (k + 1)) && !(k < rank && list[k] == r)) {
        a.rec_T[r] = kNever;
        rp_mark_rec_targets(a, r);
      }
    }
  } else if (a.c.mark_moved) {
    for (uint32_t k = rank; k < old_n; ++k) rp_mark_rec_targets(a, list[k]);   // ranked before, not now (or marked already)
  }
  a.sub_n[base] = rank;
  // an excursion that does not fit stops the super-step behind its last ranked record
  if (truncated) atomicMin(&c.smax_cut, ((unsigned long long)base << kRankBits) | (rank + 1));
}

RP_FN void rp_phase_mincut(const Args& a, uint32_t tid) {
  Ctl& c = *a.ctl;
  if (tid >= c.n_threads) return;
  atomicMin(&c.first_change, a.rec_T[a.cp[tid]]);
}

RP_FN void rp_phase_raise_fold(const Args& a, uint32_t tid) {
  Ctl& c = *a.ctl;
  if (tid >= c.a_tgt) return;
  rp_fold_raise(a, tid);
}
RP_FN void rp_phase_commit_fold(const Args& a, uint32_t tid) {
  Ctl& c = *a.ctl;
  if (tid >= c.a_tgt) return;
  rp_fold(a, tid, c.cut, true);
}

// SCAN counts / applies.  PH_RANK: item = base record i, count = its committed records (itself included), apply =
// dense pop rank of base record i.  PH_PUSH: item = j-th committed record in pop order, count[k] = its entries for
// bucket push_b[k], apply = they are written behind the bucket's tail in LUT order.
// the committed pushes of record r, one LUT index at a time: f(lut, component k of the current PH_PUSH pass)
template <class F>
RP_FN void rp_for_pushes(const Args& a, uint32_t r, const F& f) {
  const Ctl& c = *a.ctl;
  for (uint32_t w = 0; w < 7; ++w) {
    uint32_t word = a.rec_push[r * 7 + w];
    for (uint32_t j = 0; word != 0u && j < 4; ++j, word >>= 8) {
      const uint32_t v = word & 0xFF;
      if (v == 0u) continue;
      const uint32_t lut = w * 4 + j;
      uint32_t k = 0;
      while (k < c.push_n && c.push_b[k] != v - 1u) ++k;
      if (k == c.push_n) continue;                            // a bucket of another pass
      // an entry that was popped inside this super-step is not queued
      const uint32_t kid = a.rec_kid[(size_t)r * 26 + lut];
      if (kid != 0u) {
        const uint32_t kr = kid - 1;
        if (rp_meta_live(a.rec_meta[kr]) && a.rec_T[kr] < c.cut) continue;
      }
      f(lut, k);
    }
  }
}
RP_FN Cnt4 rp_scan_count(const Args& a, uint32_t i) {
  const Ctl& c = *a.ctl;
  Cnt4 n{};
  if (c.phase == PH_RANK) {
    const unsigned long long T0 = (unsigned long long)i << kRankBits;
    if (T0 < c.cut) {
      n.v[0] = 1 + a.sub_n[i];
      if ((c.cut >> kRankBits) == i) n.v[0] = (uint32_t)(c.cut & kRankMask);  // ranks in front of the cut's, plus the base record
    }
  } else {
    rp_for_pushes(a, a.ord[i], [&](uint32_t, uint32_t k) { ++n.v[k]; });
  }
  return n;
}
RP_FN void rp_scan_apply(const Args& a, uint32_t i, const Cnt4& prefix) {
  const Ctl& c = *a.ctl;
  if (c.phase == PH_RANK) {
    a.off0[i] = prefix.v[0];
    return;
  }
  const uint32_t r = a.ord[i];
  const uint32_t gid = a.rec_vox[r];
  Cnt4 pos = prefix;   // (LUT indices come in ascending order: the entries of a bucket land in LUT order)
  rp_for_pushes(a, r, [&](uint32_t lut, uint32_t k) {
    rp_queue_store(a, (int)c.push_b[k], c.tail[c.push_b[k]] + pos.v[k]++, rp_neighbour(a, gid, (int)lut));
  });
}
RP_FN void rp_phase_rank_write(const Args& a, uint32_t tid) {
  Ctl& c = *a.ctl;
  if (tid >= c.n_rec) return;
  const uint32_t m = a.rec_meta[tid];
  const unsigned long long T = a.rec_T[tid];
  if (!rp_meta_live(m) || !(T < c.cut)) return;
  const uint32_t base = (uint32_t)(T >> kRankBits);
  a.ord[a.off0[base] + (uint32_t)(T & kRankMask)] = tid;
}
// is record r one of ord[0 .. n_commit)?  (what rp_phase_rank_write files, or the identity when no dense ranking ran)
RP_FN bool rp_rec_committed(const Args& a, uint32_t r) {
  const Ctl& c = *a.ctl;
  if (c.ord_identity) return r < c.n_commit;
  return rp_meta_live(a.rec_meta[r]) && a.rec_T[r] < c.cut;
}
// (skip_committed: the last PH_PUSH pass of the device cleans up in the same launch — the thread that files a committed record's
// pushes clears that record's words itself when it has read them, everybody else's are cleared here)
RP_FN void rp_phase_cleanup(const Args& a, uint32_t tid, bool skip_committed = false) {
  Ctl& c = *a.ctl;
  if (tid < c.a_tgt) {
    const uint32_t gid = a.tgt_gid[tid];
    if (gid != kNone) a.vox2tgt[gid] = 0;
    a.tgt_gid[tid] = kNone;   // (an id nobody takes in the next super-step is a hole)
    a.tgt_cnt[tid] = 0;
    a.tgt_dirty[tid] = 0;
  }
  if (tid < c.n_rec && !(skip_committed && rp_rec_committed(a, tid))) {
    for (int k = 0; k < 26; ++k) a.rec_kid[(size_t)tid * 26 + k] = 0;
    for (int k = 0; k < 7; ++k) a.rec_push[tid * 7 + k] = 0;
    a.rec_poison[tid] = 0;
  }
}

// ------------------------------------------------------------------------------------------------------------
// control: one thread, after every phase
// ------------------------------------------------------------------------------------------------------------
RP_FN void rp_stop(Ctl& c) { c.phase = PH_DONE; c.done = 1; c.n_threads = 0; }

RP_FN void rp_begin_superstep(const Args& a) {
  Ctl& c = *a.ctl;
  int b = a.c.num_buckets;   // processRaiseSet runs to the end before processOpenSet starts (:293-298)
  c.raise = 1;
  if (c.head[b] == c.tail[b]) {
    c.raise = 0;
    b = 0;
    while (b < a.c.num_buckets && c.head[b] == c.tail[b]) ++b;
    if (b == a.c.num_buckets) { rp_stop(c); return; }
  }
  c.bucket = (uint32_t)b;
  uint32_t K = c.tail[b] - c.head[b];
  if (c.k_cur[b] == 0 || c.k_cur[b] > a.c.kmax) c.k_cur[b] = a.c.kmax;
  if (K > c.k_cur[b]) K = c.k_cur[b];
  c.K = K;
  c.base_head = c.head[b];
  c.n_rec = K;
  c.n_tgt = 0;
  for (uint32_t k = 0; k < kTgtShards; ++k) c.tgt_n[k].v = 0;
  c.iter = 0;
  c.read = 1;            // PLACE_BASE marks into list 0
  c.n_dirty[0] = c.n_dirty[1] = 0;
  for (uint32_t k = 0; k < kShards; ++k) c.chg_n[k].v = c.born_n[k].v = 0;
  c.n_sd = c.n_cp = 0;
  c.cap_stop = 0;
  c.k_limit = kNone;
  c.first_change = kNever;
  c.smax_cut = kNever;
  c.cut = kNever;
  *a.sub_slots_used = 0;
  if (c.raise) ++c.st_raise_steps; else ++c.st_supersteps;
  c.phase = PH_PLACE_BASE;
  c.n_threads = K * 27;
}

// The FIRST base record could not be heard by one of its targets: event lists fill up first come, first served, and with
// multi_queue a bucket holds the same voxels many times over, so a prefix of the bucket may well fit where all of it does
// not.  Nothing has been committed; the super-step is taken apart and started again with a quarter of the records
// (PH_CLEANUP sees cut == 0).  A single record that does not fit is the one hard case.
RP_FN void rp_retry_smaller(const Args& a) {
  Ctl& c = *a.ctl;
  if (c.K <= 1) { c.error |= 16u; rp_stop(c); return; }
  c.a_tgt = RP_LD(c.n_tgt);
  if (c.a_tgt > a.tgt_cap) c.a_tgt = a.tgt_cap;
  c.cut = 0;
  c.n_commit = 0;
  ++c.st_retries;
  c.phase = PH_CLEANUP;
  c.n_threads = c.a_tgt > c.n_rec ? c.a_tgt : c.n_rec;
}

RP_FN void rp_start_commit(const Args& a) {
  Ctl& c = *a.ctl;
  const unsigned long long first_change = RP_LD64(c.first_change), smax_cut = RP_LD64(c.smax_cut);
  const uint32_t k_limit = RP_LD(c.k_limit);
  unsigned long long cut = first_change < smax_cut ? first_change : smax_cut;
  if (k_limit != kNone) {
    const unsigned long long kl = (unsigned long long)k_limit << kRankBits;
    if (kl < cut) cut = kl;
  }
  if (cut != kNever) { if (smax_cut <= first_change) ++c.st_cut_smax; else ++c.st_cut_iters; }
  if (cut == 0) {
    if (k_limit == 0) { rp_retry_smaller(a); return; }
    c.error |= 8u;
    rp_stop(c);
    return;
  }
  c.cut = cut;
  c.a_tgt = RP_LD(c.n_tgt);
  if (c.a_tgt > a.tgt_cap) c.a_tgt = a.tgt_cap;
  for (int k = 0; k < a.c.num_buckets; ++k) c.push_cnt[k] = 0;
  c.phase = PH_COMMIT_FOLD;
  c.n_threads = c.a_tgt;
}

// the next (up to kScanC) buckets that receive entries from this commit, chunks reserved for the most they can get
RP_FN void rp_next_push_pass(const Args& a) {
  Ctl& c = *a.ctl;
  c.push_n = 0;
  int b = (int)c.push_next;
  const int nq = a.c.num_buckets + (c.raise ? 1 : 0);
  for (; b < nq && c.push_n < (uint32_t)kScanC; ++b) {
    const uint32_t bound = RP_LD(c.push_cnt[b]);
    if (bound == 0) continue;
    if (!rp_queue_reserve(a, b, bound)) { rp_stop(c); return; }
    c.push_b[c.push_n++] = (uint32_t)b;
  }
  c.push_next = (uint32_t)b;
  if (c.push_n == 0) {
    c.phase = PH_CLEANUP;
    c.n_threads = c.a_tgt > c.n_rec ? c.a_tgt : c.n_rec;
    return;
  }
  c.phase = PH_PUSH;
  c.scan_n = c.n_commit;
  c.n_threads = c.n_commit;
  // the last pass cleans up behind itself: one launch less per super-step
  c.push_last = 1;
  for (int b2 = b; b2 < nq; ++b2)
    if (RP_LD(c.push_cnt[b2]) != 0) { c.push_last = 0; break; }
  if (c.push_last) {
    const uint32_t items = c.a_tgt > c.n_rec ? c.a_tgt : c.n_rec;
    if (items > c.n_threads) c.n_threads = items;
  }
}

RP_FN void rp_control(const Args& a) {
  Ctl& c = *a.ctl;
  if (a.c.stats) {   // (per-step diagnostics: three read-modify-writes of the control block per step otherwise — 0.2 us of every launch)
    ++c.st_steps;
    ++c.st_phase_steps[c.phase & 15];
    c.st_phase_threads[c.phase & 15] += c.n_threads;
  }
  if (RP_LD(c.error)) { rp_stop(c); return; }
  switch (c.phase) {
    case PH_BEGIN:
      rp_begin_superstep(a);
      break;
    case PH_PLACE_BASE:
      if (a.c.tgt_shards > 1u) {   // the ids in use end at the longest shard's last one
        uint32_t top = RP_LD(c.n_tgt);
        for (uint32_t k = 0; k < a.c.tgt_shards && k < kTgtShards; ++k) {
          const uint32_t nk = RP_LD(c.tgt_n[k].v);
          if (nk && (nk - 1u) * a.c.tgt_shards + k + 1u > top) top = (nk - 1u) * a.c.tgt_shards + k + 1u;
        }
        c.n_tgt = top;
      }
      if (RP_LD(c.k_limit) == 0) { rp_retry_smaller(a); break; }
      if (c.raise) {
        c.a_tgt = RP_LD(c.n_tgt);
        if (c.a_tgt > a.tgt_cap) c.a_tgt = a.tgt_cap;
        const uint32_t kl = RP_LD(c.k_limit);
        if (kl != kNone) c.cut = (unsigned long long)kl << kRankBits;   // an event list overflowed: the records behind wait
        for (int k = 0; k <= a.c.num_buckets; ++k) c.push_cnt[k] = 0;
        c.phase = PH_RAISE_FOLD;
        c.n_threads = c.a_tgt;
        break;
      }
      // fall through
    case PH_SIM:
    case PH_APPLY: {
      if (c.phase == PH_APPLY) c.n_cp = c.a_chg + c.a_born;   // PH_APPLY leaves one change point per item (Args::cp[item])
      const uint32_t n_cp = c.n_cp;
      if (c.phase == PH_APPLY) {
        if (!c.cap_stop) {
          c.n_rec += c.a_born;
          c.st_exc += c.a_born;
        }
        for (uint32_t k = 0; k < kShards; ++k) c.chg_n[k].v = c.born_n[k].v = 0;
        const uint32_t n_sd = RP_LD(c.n_sd);
        if (n_sd) { c.phase = PH_SIM; c.n_threads = n_sd; break; }
      }
      if (c.phase == PH_SIM) c.n_sd = 0;
      // next iteration, or stop
      if (c.phase != PH_PLACE_BASE && n_cp == 0) { rp_start_commit(a); break; }   // (cannot happen: APPLY always leaves a change point)
      if (c.iter >= a.c.max_iters || c.cap_stop) { c.phase = PH_MINCUT; c.n_threads = n_cp; break; }
      c.n_cp = 0;
      c.read = 1u - c.read;
      c.n_dirty[1u - c.read] = 0;
      ++c.iter;
      ++c.st_iters;
      c.n_threads = RP_LD(c.n_dirty[c.read]);
      c.fold_all = 0;
      if (c.phase == PH_PLACE_BASE && a.c.fold_all) {
        c.fold_all = 1;
        c.n_threads = RP_LD(c.n_tgt);
        if (c.n_threads > a.tgt_cap) c.n_threads = a.tgt_cap;
      }
      c.st_folds += c.n_threads;
      c.phase = PH_FOLD;
      break;
    }
    case PH_FOLD: {
      c.a_chg = c.a_born = 0;
      for (uint32_t k = 0; k < kShards; ++k) {
        uint32_t nc = RP_LD(c.chg_n[k].v), nbn = RP_LD(c.born_n[k].v);
        if (nc > a.rec_cap) nc = a.rec_cap;      // (what a shard could not hold was dealt with where it was found)
        if (nbn > a.rec_cap) nbn = a.rec_cap;
        c.chg_pre[k] = c.a_chg;
        c.born_pre[k] = c.a_born;
        c.a_chg += nc;
        c.a_born += nbn;
      }
      c.chg_pre[kShards] = c.a_chg;
      c.born_pre[kShards] = c.a_born;
      if (c.a_chg + c.a_born > 4u * a.rec_cap) { c.error |= 1u; rp_stop(c); break; }   // (Args::cp holds 4 * rec_cap change points: a record is listed at most twice, born at most rec_cap)
      if (c.a_chg == 0 && c.a_born == 0) { c.n_cp = 0; rp_start_commit(a); break; }   // fixed point
      if (c.n_rec + c.a_born > a.rec_cap) c.cap_stop = 1;   // PH_APPLY applies the changes, notes the pushers of the births it cannot make, and the prefix commits
      c.phase = PH_APPLY;
      c.n_threads = (c.a_chg + c.a_born) * 27;
      break;
    }
    case PH_MINCUT:
      rp_start_commit(a);
      break;
    case PH_RAISE_FOLD:
      c.n_commit = c.cut == kNever ? c.K : (uint32_t)(c.cut >> kRankBits);
      c.push_next = 0;
      rp_next_push_pass(a);
      break;
    case PH_COMMIT_FOLD:
      if (c.n_rec == c.K) {
        // no excursion records: the committed records are the base records in front of the cut, in FIFO order — which is
        // what ord[] holds since PLACE_BASE (ord[r] = r); the dense ranking and its write pass have nothing to add
        unsigned long long nb = c.cut >> kRankBits;
        c.n_commit = (c.cut == kNever || nb > c.K) ? c.K : (uint32_t)nb;
        c.push_next = 0;
        c.ord_identity = 1;
        rp_next_push_pass(a);
        break;
      }
      c.ord_identity = 0;
      c.phase = PH_RANK;
      c.n_threads = c.K;
      break;
    case PH_RANK:
      c.n_commit = RP_LD(c.scan_tot[0]);
      c.phase = PH_RANK_WRITE;
      c.n_threads = c.n_rec;
      break;
    case PH_RANK_WRITE:
      c.push_next = 0;
      rp_next_push_pass(a);
      break;
    case PH_PUSH:
      for (uint32_t k = 0; k < c.push_n; ++k) c.tail[c.push_b[k]] += RP_LD(c.scan_tot[k]);
      if (!c.push_last) {
        rp_next_push_pass(a);
        break;
      }
      c.push_last = 0;
      // fall through: the pass cleaned up
    case PH_CLEANUP: {
      // the committed base records leave their FIFO
      uint32_t nb = (uint32_t)(c.cut >> kRankBits);
      if (c.cut == kNever || nb > c.K) nb = c.K;
      else if ((c.cut & kRankMask) != 0) nb += 1;   // the cut lies inside base record nb's excursion: it has popped
      c.head[c.bucket] += nb;
      if (c.raise) c.st_raise_pops += c.n_commit; else c.st_pops += c.n_commit;
      // a cut throws the work behind it away: take about as much as got through next time, ramp up after clean steps
      if (c.cut == 0) c.k_cur[c.bucket] = c.K / 4 > 1 ? c.K / 4 : 1;   // rp_retry_smaller
      else if (c.raise) c.k_cur[c.bucket] = c.k_cur[c.bucket] * 4 > a.c.kmax ? a.c.kmax : c.k_cur[c.bucket] * 4;   // (raise super-steps only cut at a full event list)
      else if (c.cut != kNever) c.k_cur[c.bucket] = nb * a.c.cut_mult > 32 ? nb * a.c.cut_mult : 32;
      else c.k_cur[c.bucket] = c.k_cur[c.bucket] * a.c.ramp_mult > a.c.kmax ? a.c.kmax : c.k_cur[c.bucket] * a.c.ramp_mult;
      rp_begin_superstep(a);
      break;
    }
    default:
      rp_stop(c);
      break;
  }
}

}  // namespace rp
