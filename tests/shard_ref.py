"""Serial restatement of the ray-bundle shard + merge of voxblox_amd.multi_gpu / libvbx_shard.so with the CPU
oracle (test infrastructure): per time step every rank's shards go, one after the other, into ONE zeroed delta
map of that rank (a fresh oracle map and a fresh integrator per shard call = the per-frame ApproxHashSet reset
of FastTsdfIntegrator, tsdf_integrator.cc:564-569), the deltas' weighted sums (w*d, w, w*rgba) of a block are
added in rank order (float32, the owner's row order) and merged into the persistent voxel with
mergeVoxelAIntoVoxelB (voxel_utils.cc:10-22; Block::mergeBlock block_inl.h:112-129)."""
import numpy as np

from test_multi_gpu_gloo import merge_A_into_B


def serial_shard_merge(O, voxel, kind, ocfg, steps, vps=16, deltas_per_rank=1):
    """steps[k][rank] = [(pos, quat, points, colors), ...]  ->  {BlockIndex: (d, w, rgba)} of the merged map.
    deltas_per_rank: shard i of a rank goes into that rank's delta i % deltas_per_rank (ShardedTsdfMap's rule);
    1 = every shard of a rank into one delta map, one after the other."""
    nv = vps ** 3
    ref = {}
    for per_rank in steps:
        sums = {}    # BlockIndex -> six float32 planes, rows added in (rank, delta) order
        order = []
        for shards in per_rank:
            if not shards:
                continue
            nd = max(1, min(deltas_per_rank, len(shards)))
            for u in range(nd):
                m = O.OracleMap(voxel, vps)
                for pos, quat, pts, col in [sh for i, sh in enumerate(shards) if i % deltas_per_rank == u]:
                    O.lib().orc_fast_reset_counter_set(0)
                    m.tsdf_integrator(kind, ocfg).integrate(pos, quat, pts, col)
                # the rows of one sender arrive in (delta, z,y,x) order; the order across blocks does not matter for the sums
                for key, (d, w, c, _) in m.tsdf_dict().items():
                    sA = np.stack([w * d, w] + [w * c[:, ch].astype(np.float32) for ch in range(4)]).astype(np.float32)
                    if key in sums:
                        sums[key] = (sums[key] + sA).astype(np.float32)
                    else:
                        sums[key] = sA
                        order.append(key)
                del m
        for key in order:
            sA = sums[key]
            if not (sA[1] > 0).any():
                continue
            dB, wB, cB = ref.get(key, (np.zeros(nv, np.float32), np.zeros(nv, np.float32), np.zeros((nv, 4), np.uint8)))
            ref[key] = merge_A_into_B(sA, dB, wB, cB)
    return ref


def assert_merged_equal(got, ref, d_tol=1e-5):
    """got: {key: (d, w, rgba, updated)} from the HIP map(s); ref from serial_shard_merge.  Distances and weights
    within d_tol (relative for the weights), colours +-1 LSB (one rounding after the sum vs one per pairwise blend,
    SURVEY 8(e)), block sets and observed masks equal."""
    assert set(got) == set(ref), (len(got), len(ref), list(set(got) ^ set(ref))[:5])
    worst_d = worst_w = 0.0
    for key in ref:
        gd, gw, gc = got[key][0], got[key][1], got[key][2]
        rd, rw, rc = ref[key]
        assert np.array_equal(gw > 0, rw > 0), key
        on = rw > 0
        if on.any():
            worst_d = max(worst_d, float(np.abs(gd[on] - rd[on]).max()))
            worst_w = max(worst_w, float((np.abs(gw[on] - rw[on]) / rw[on]).max()))
        assert np.abs(gc.astype(np.int32) - rc.astype(np.int32)).max() <= 1, key
    assert worst_d <= d_tol and worst_w <= d_tol, (worst_d, worst_w)
    return worst_d, worst_w
