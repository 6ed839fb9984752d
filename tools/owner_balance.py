"""Owner balance of the multi-GPU exchange, WITHOUT hardware (CPU only; the touched-block sets come from the oracle = what the
device integrates bit for bit): for BASELINE configs[4] (4 sensors, 0.02 m, bench.py's dealing of ray bundles to ranks) and for
the weak-scaling stream (configs[1] on every rank) at world 2 / 4 / 8 — blocks and bytes every rank SENDS per step, blocks and bytes
every OWNER receives and folds per step (owner = multi_gpu.owner_of(BlockIndex) mod world), their max / mean imbalance, and the
time the busiest xGMI endpoint needs at 153 GB/s per link (all-to-all-v: every pair of ranks has a link of its own, so the bound
is the busiest single (sender, owner) pair and the busiest rank's total over its 7 links).
usage: python tools/owner_balance.py [STEPS]  -> JSON on stdout (profiles/r06_owner_balance.json)"""
import json
import os
import sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import oracle_py as O  # noqa: E402
from voxblox_amd import multi_gpu, scenes  # noqa: E402

ROW_BYTES = 3 * 4096 * 4          # a row of the exchange = the delta block: distance, weight, colour planes (include/vbx_shard.h)
LINK_GBPS = 153.0                 # one xGMI link, MI355X_MICROARCH.md


def touched_blocks(voxel, pose, pts, col):
    O.lib().orc_fast_reset_counter_set(0)
    m = O.OracleMap(voxel, 16)
    it = m.tsdf_integrator("fast", O.tsdf_cfg(default_truncation_distance=4 * voxel, integrator_threads=8))
    it.integrate(pose[0], pose[1], pts, col)
    # (thread count does not matter for WHICH blocks a cloud touches to within a few blocks; the exact sets are the 1-thread ones,
    # taken when VBX_BALANCE_EXACT=1 — 8x slower)
    return m.block_indices(0)


def balance(units_by_rank, world):
    """units_by_rank[r] = list of (n,3) block index arrays, one per ray bundle of rank r in one step"""
    sent = np.zeros(world, np.int64)
    recv = np.zeros(world, np.int64)
    pair = np.zeros((world, world), np.int64)
    owned = set()
    for r, units in enumerate(units_by_rank):
        for keys in units:
            own = multi_gpu.owner_of(keys, world)
            sent[r] += len(keys)
            for o in range(world):
                c = int((own == o).sum())
                pair[r, o] += c
                recv[o] += c
            owned.update(tuple(int(v) for v in k) for k in keys)
    off = pair.copy()
    np.fill_diagonal(off, 0)      # rows a rank owns itself do not cross a link
    link_busiest = int(off.max())
    endpoint_busiest = int(max(off.sum(1).max(), off.sum(0).max()))
    return {"sent_blocks": sent.tolist(), "received_blocks": recv.tolist(), "distinct_blocks": len(owned),
            "sent_MB": [round(x * ROW_BYTES / 1e6, 1) for x in sent], "received_MB": [round(x * ROW_BYTES / 1e6, 1) for x in recv],
            "receive_imbalance_max_over_mean": round(float(recv.max() / max(recv.mean(), 1e-9)), 3),
            "send_imbalance_max_over_mean": round(float(sent.max() / max(sent.mean(), 1e-9)), 3),
            "busiest_pair_MB": round(link_busiest * ROW_BYTES / 1e6, 1),
            "busiest_pair_ms_at_one_xgmi_link": round(link_busiest * ROW_BYTES / (LINK_GBPS * 1e9) * 1e3, 3),
            "busiest_endpoint_off_rank_MB": round(endpoint_busiest * ROW_BYTES / 1e6, 1),
            "busiest_endpoint_ms_over_7_links": round(endpoint_busiest * ROW_BYTES / (7 * LINK_GBPS * 1e9) * 1e3, 3)}


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    out = {"row_bytes": ROW_BYTES, "owner": "multi_gpu.owner_of(BlockIndex) mod world (the same function in libvbx_shard.so)",
           "configs4": {}, "stream": {}}
    # ---- configs[4]: 4 sensors, 0.02 m; bundles dealt as bench.py deals them
    cache = {}
    for world in (2, 4, 8):
        per_step = []
        for step in range(steps):
            units_by_rank = []
            for r in range(world):
                units = []
                for s, b, bands in multi_gpu.deal_sensor_units(world, bands=max(1, (world + 3) // 4))[r]:
                    key = (s, step, b, bands)
                    if key not in cache:
                        pose, pts, col = scenes.room_sensor_frame(s, step)
                        lo, hi = multi_gpu.band_of(pts.shape[0], b, bands)
                        cache[key] = touched_blocks(0.02, pose, pts[lo:hi], col[lo:hi])
                    units.append(cache[key])
                units_by_rank.append(units)
            per_step.append(balance(units_by_rank, world))
        out["configs4"][str(world)] = per_step[-1]
        out["configs4"][str(world)]["steps_looked_at"] = steps
        out["configs4"][str(world)]["bundles_per_sensor"] = max(1, (world + 3) // 4)
    # ---- weak-scaling stream: one 640x480 sensor per rank, 0.05 m (bench.stream_frames)
    import bench
    for world in (2, 4, 8):
        units_by_rank = []
        for r in range(world):
            pose, pts, col = bench.stream_frames("room", r, 8, world)[7]
            units_by_rank.append([touched_blocks(0.05, pose, pts, col)])
        out["stream"][str(world)] = balance(units_by_rank, world)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
