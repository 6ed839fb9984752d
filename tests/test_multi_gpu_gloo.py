"""CPU, world_size 2, gloo: the N>1 path of voxblox_amd.multi_gpu (owner grouping, sparse
all-to-all of the touched blocks' sums, owner-side sum of duplicate rows, owner fold) with an
oracle-backed backend, checked against
the same shard + merge done serially with the reference's mergeVoxelAIntoVoxelB
(voxel_utils.cc:10-22).  The HIP kernels behind the same protocol are covered by
tests/test_gpu_multi_merge.py."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleBackend:
    """The backend protocol of multi_gpu.ShardedTsdfMap on top of the CPU oracle (numpy)."""

    def __init__(self, O, voxel, vps=16):
        import torch
        self.O, self.voxel, self.vps = O, voxel, vps
        self.nvox = vps ** 3
        self.device = torch.device("cpu")
        self._torch = torch
        self.m = O.OracleMap(voxel, vps)
        self.it = None

    def clear(self):
        self.m.clear(0)

    def integrate(self, kind, cfg, pos, quat, points, colors, n_points=None):
        self.O.lib().orc_fast_reset_counter_set(0)
        it = self.m.tsdf_integrator(kind, cfg)
        it.integrate(pos, quat, points, colors)

    def block_indices(self):
        return self.m.block_indices(0)

    def zeros(self, shape):
        return self._torch.zeros(shape, dtype=self._torch.float32)

    def export_sums(self, keys, out_view):
        # a row = the delta block itself: distance, weight, colour bytes (vbx_blocks_export_sums, 3 planes)
        o = out_view.numpy()
        for i, k in enumerate(keys):
            blk = self.m.tsdf_block(k)
            if blk is None:
                continue
            d, w, c, _ = blk
            o[i, 0] = d
            o[i, 1] = w
            o[i, 2] = np.ascontiguousarray(c, np.uint8).view(np.uint32).reshape(-1).view(np.float32)

    def merge_sums(self, keys, sums, apply_caps, trunc, max_weight):
        # the owner forms the six weighted sums per row and adds the rows of one block in row order (vbx_blocks_merge_sums)
        raw = sums.numpy()
        rows = np.zeros((raw.shape[0], 6, raw.shape[2]), np.float32)
        for i in range(raw.shape[0]):
            w = raw[i, 1]
            c = np.ascontiguousarray(raw[i, 2]).view(np.uint32).view(np.uint8).reshape(-1, 4)
            rows[i, 0] = w * raw[i, 0]
            rows[i, 1] = w
            for ch in range(4):
                rows[i, 2 + ch] = w * c[:, ch].astype(np.float32)
        first = {}
        acc = []
        for i, k in enumerate(keys):
            kk = tuple(int(v) for v in k)
            if kk in first:
                acc[first[kk]][1] = acc[first[kk]][1] + rows[i]
            else:
                first[kk] = len(acc)
                acc.append([k, rows[i].copy()])
        keys = [a[0] for a in acc]
        s = [a[1] for a in acc]
        for i, k in enumerate(keys):
            wA = s[i][1]
            if not (wA > 0).any():
                continue
            blk = self.m.tsdf_block(k)
            if blk is None:
                n = self.nvox
                dB, wB, cB = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros((n, 4), np.uint8)
            else:
                dB, wB, cB, _ = blk
            dB, wB, cB = merge_A_into_B(s[i], dB, wB, cB)
            self.m.tsdf_block_set(k, dB, wB, cB, 7)


def merge_A_into_B(sA, dB, wB, cB):
    """mergeVoxelAIntoVoxelB with A given as sums (w*d, w, w*rgba)."""
    wA = sA[1].astype(np.float32)
    on = wA > 0
    with np.errstate(divide="ignore", invalid="ignore"):
        dA = np.where(on, sA[0] / wA, 0).astype(np.float32)
        cA = np.stack([np.where(on, np.round(sA[2 + ch] / wA), 0) for ch in range(4)], 1).astype(np.float32)
        cw = (wA + wB).astype(np.float32)
        d = np.where(on & (cw > 0), ((dA * wA + dB * wB) / cw).astype(np.float32), dB)
        f1 = (wA / cw).astype(np.float32)
        f2 = (wB / cw).astype(np.float32)
        col = np.where((on & (cw > 0))[:, None],
                       np.round(cA * f1[:, None] + cB.astype(np.float32) * f2[:, None]), cB).astype(np.uint8)
    w = np.where(on & (cw > 0), cw, wB).astype(np.float32)
    return d.astype(np.float32), w, col


def _worker(rank, world, port, out_q, pipelined=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import oracle_py as O
    from voxblox_amd import multi_gpu, scenes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    voxel = 0.1
    cfg = O.tsdf_cfg(default_truncation_distance=4 * voxel, integrator_threads=1)
    if pipelined:  # double-buffered deltas, exchange on a worker thread (all collectives issued there)
        sm = multi_gpu.PipelinedShardedTsdfMap(OracleBackend(O, voxel), [OracleBackend(O, voxel), OracleBackend(O, voxel)],
                                               rank, world, dist)
    else:
        sm = multi_gpu.ShardedTsdfMap(OracleBackend(O, voxel), OracleBackend(O, voxel), rank, world, dist)
    for k in range(3):  # every rank: its own band of the same frame (ray-bundle sharding)
        pose, pts, col = scenes.room_frame(7 * k, 100, f=40.0, width=80, height=60)
        n = pts.shape[0]
        lo, hi = rank * n // world, (rank + 1) * n // world
        sm.integrate_shard("simple", cfg, pose[0], pose[1], pts[lo:hi], col[lo:hi])
    if pipelined:
        sm.close()
    owned = {tuple(int(v) for v in i): sm.p.m.tsdf_block(i) for i in sm.p.block_indices()}
    out_q.put((rank, owned, sm.last))
    dist.barrier()
    dist.destroy_process_group()


def test_owner_grouping_is_deterministic():
    sys.path.insert(0, ROOT)
    from voxblox_amd import multi_gpu
    rng = np.random.RandomState(0)
    a = np.unique(rng.randint(-20, 20, (90, 3)).astype(np.int32), axis=0)
    g1, c1 = multi_gpu.group_by_owner(a, 4)
    g2, c2 = multi_gpu.group_by_owner(a[rng.permutation(a.shape[0])], 4)      # input order does not matter
    assert np.array_equal(g1, g2) and np.array_equal(c1, c2) and int(c1.sum()) == a.shape[0]
    off = 0
    for r in range(4):
        g = g1[off:off + int(c1[r])]
        assert np.all(multi_gpu.owner_of(g, 4) == r)
        assert np.array_equal(g, multi_gpu._sort_rows_zyx(g))      # (z,y,x) order inside a group
        off += int(c1[r])
    g0, c0 = multi_gpu.group_by_owner(np.zeros((0, 3), np.int32), 4)
    assert g0.shape == (0, 3) and c0.tolist() == [0, 0, 0, 0]


@pytest.mark.parametrize("pipelined", [False, True])
def test_two_rank_shard_and_merge_matches_serial_reference_merge(oracle, pipelined):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, pipelined)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    merged = {}
    for rank, owned, last in results:
        assert not (set(owned) & set(merged)), "a block is owned by two ranks"
        merged.update(owned)
        # sparse exchange: a rank sends exactly the blocks its delta touched, 96 KiB of sums each
        assert last["sent_blocks"] > 0 and last["payload_bytes"] == last["sent_blocks"] * 3 * 4096 * 4
        assert last["received_blocks"] >= last["owned_blocks"] > 0

    # serial restatement: per frame, per rank delta (fresh map), merged in rank order with the
    # reference's mergeVoxelAIntoVoxelB
    from voxblox_amd import multi_gpu, scenes
    voxel = 0.1
    cfg = oracle.tsdf_cfg(default_truncation_distance=4 * voxel, integrator_threads=1)
    ref = {}
    for k in range(3):
        pose, pts, col = scenes.room_frame(7 * k, 100, f=40.0, width=80, height=60)
        n = pts.shape[0]
        for rank in range(2):
            lo, hi = rank * n // 2, (rank + 1) * n // 2
            m = oracle.OracleMap(voxel, 16)
            m.tsdf_integrator("simple", cfg).integrate(pose[0], pose[1], pts[lo:hi], col[lo:hi])
            for key, (d, w, c, _) in m.tsdf_dict().items():
                if not (w > 0).any():
                    continue
                dB, wB, cB = ref.get(key, (np.zeros(4096, np.float32), np.zeros(4096, np.float32),
                                           np.zeros((4096, 4), np.uint8)))
                sA = np.stack([w * d, w] + [w * c[:, ch].astype(np.float32) for ch in range(4)])
                ref[key] = merge_A_into_B(sA, dB, wB, cB)
    assert set(merged) == set(ref), (len(merged), len(ref))
    for key in ref:
        gd, gw, gc, _ = merged[key]
        rd, rw, rc = ref[key]
        assert np.array_equal(gw > 0, rw > 0)
        assert np.allclose(gw, rw, rtol=1e-5, atol=1e-6)
        assert np.abs(gd - rd).max() <= 1e-5
        # one rounding after the sum vs a rounding at every pairwise blend: +-1 LSB
        assert np.abs(gc.astype(np.int32) - rc.astype(np.int32)).max() <= 1
    # owners partition the union exactly as owner_of says
    for rank, owned, _ in results:
        keys = np.array(list(owned.keys()), np.int32).reshape(-1, 3)
        if keys.shape[0]:
            assert np.all(multi_gpu.owner_of(keys, 2) == rank)


def _worker_multi_delta(rank, world, port, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import oracle_py as O
    from voxblox_amd import multi_gpu, scenes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    voxel = 0.1
    cfg = O.tsdf_cfg(default_truncation_distance=4 * voxel, integrator_threads=1)
    # two ray shards per rank and step, a delta map each, integrated concurrently; pipelined exchange
    sm = multi_gpu.PipelinedShardedTsdfMap(OracleBackend(O, voxel), [[OracleBackend(O, voxel) for _ in range(2)] for _ in range(2)],
                                           rank, world, dist)
    for k in range(3):
        pose, pts, col = scenes.room_frame(7 * k, 100, f=40.0, width=80, height=60)
        n = pts.shape[0]
        q = n // (2 * world)
        shards = [(pose[0], pose[1], pts[(2 * rank + j) * q:(2 * rank + j + 1) * q], col[(2 * rank + j) * q:(2 * rank + j + 1) * q], None)
                  for j in range(2)]
        sm.integrate_shards("merged", cfg, shards)
    sm.close()
    owned = {tuple(int(v) for v in i): sm.p.m.tsdf_block(i) for i in sm.p.block_indices()}
    out_q.put((rank, owned))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_two_concurrent_shards_each_matches_serial_merge(oracle):
    """ShardedTsdfMap with one delta map per ray shard (the layout bench.py runs for configs[4]): shards of a rank
    integrated concurrently, rows of a block added in (rank, delta) = global shard order at the owner."""
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from shard_ref import assert_merged_equal, serial_shard_merge
    from voxblox_amd import scenes
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_multi_delta, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    merged = {}
    for rank, owned in results:
        assert not (set(owned) & set(merged))
        merged.update(owned)
    voxel = 0.1
    cfg = oracle.tsdf_cfg(default_truncation_distance=4 * voxel, integrator_threads=1)
    steps = []
    for k in range(3):
        pose, pts, col = scenes.room_frame(7 * k, 100, f=40.0, width=80, height=60)
        qn = pts.shape[0] // 4
        steps.append([[(pose[0], pose[1], pts[(2 * r + j) * qn:(2 * r + j + 1) * qn], col[(2 * r + j) * qn:(2 * r + j + 1) * qn])
                       for j in range(2)] for r in range(2)])
    ref = serial_shard_merge(oracle, voxel, "merged", cfg, steps, deltas_per_rank=2)
    assert len(ref) > 20
    assert_merged_equal(merged, ref)


def test_ray_bundle_layout_is_the_same_for_every_world_size():
    """multi_gpu.deal_sensor_units: configs[4] is cut into 4 sensors x BANDS_PER_SENSOR bundles whatever the number of
    ranks (so the merged map — a function of the bundle layout — is the same on 1, 2, 4, 8 and 16 GPUs); the ranks take
    consecutive bundles, every bundle exactly once; more ranks than bundles cut finer."""
    from voxblox_amd import multi_gpu
    ref = [u for units in multi_gpu.deal_sensor_units(1) for u in units]
    assert len(ref) == 4 * multi_gpu.BANDS_PER_SENSOR and ref == [(s, b, multi_gpu.BANDS_PER_SENSOR) for s in range(4)
                                                                   for b in range(multi_gpu.BANDS_PER_SENSOR)]
    for world in (2, 4, 8, 16):
        dealt = multi_gpu.deal_sensor_units(world)
        assert len(dealt) == world and [u for units in dealt for u in units] == ref
        assert {len(units) for units in dealt} == {len(ref) // world}
    finer = multi_gpu.deal_sensor_units(32)
    assert len(finer) == 32 and all(len(u) == 1 and u[0][2] == 8 for u in finer)
    # the bands of a cloud tile it exactly
    n = 307200
    edges = [multi_gpu.band_of(n, b, 4) for b in range(4)]
    assert edges[0][0] == 0 and edges[-1][1] == n and all(edges[i][1] == edges[i + 1][0] for i in range(3))
    # explicit layouts for the parity tests
    assert [u for units in multi_gpu.deal_sensor_units(1, bands=1) for u in units] == [(s, 0, 1) for s in range(4)]


def _worker_bench_loop(rank, world, port, out_q):
    """One rank of bench.py's N > 1 timed region (bench.sharded_stream_timed: warm-up, flush, barrier-bracketed timed frames,
    MAX over ranks) over gloo with the oracle-backed map — the real world size of the driver's scaling run, no GPU."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import oracle_py as O
    import bench
    from voxblox_amd import multi_gpu, scenes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    voxel = 0.1
    cfg = O.tsdf_cfg(default_truncation_distance=4 * voxel, integrator_threads=1)
    sm = multi_gpu.PipelinedShardedTsdfMap(OracleBackend(O, voxel), [OracleBackend(O, voxel), OracleBackend(O, voxel)], rank, world, dist)
    # bench.stream_frames: every rank its own sensor along the same trajectory (weak scaling); small frames keep eight CPU ranks short
    frames = []
    for k in range(5):
        pose, pts, col = scenes.room_frame((k + rank * 100 // world) % 100, 100, f=40.0, width=80, height=60)
        frames.append((pose, pts, col, pts.shape[0]))
    n_barriers = [0]

    def barrier():
        n_barriers[0] += 1
        dist.barrier()

    warmup, total = 2, 5
    dt = bench.sharded_stream_timed(sm, frames, "fast", cfg, warmup, total, barrier, dist, world, "cpu")
    stats = dict(sm.stats)
    sm.close()
    owned = {tuple(int(v) for v in i): sm.p.m.tsdf_block(i) for i in sm.p.block_indices()}
    out_q.put((rank, dt, stats, n_barriers[0], owned))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_timed_region_at_world_size_8_over_gloo(oracle):
    """The driver's scaling run is the first time more than one RCCL rank ever executes (one GPU per development box).  What
    can be checked without the hardware is everything around the collectives at the REAL world size: eight ranks run bench.py's
    own timed region (bench.sharded_stream_timed) over gloo — nobody deadlocks (the pipelined exchange issues its collectives
    from a worker thread), every rank ends on the same MAX-over-ranks time, two barriers bracket the timed frames, every rank
    sent blocks, the owners partition the union of the blocks, and the merged map equals the serial oracle shard + merge."""
    import torch.multiprocessing as mp
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bench_loop, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    results.sort(key=lambda t: t[0])
    dts = [r[1] for r in results]
    assert all(d == dts[0] for d in dts) and dts[0] > 0, dts            # the all_reduce(MAX): one number on every rank
    merged = {}
    for rank, dt, stats, n_barriers, owned in results:
        assert n_barriers == 2                                           # barrier + sync on both sides of the timed frames
        assert stats["frames"] == 3 and stats["payload_bytes"] > 0      # the counters cover exactly the timed frames
        assert not (set(owned) & set(merged)), "a block is owned by two ranks"
        merged.update(owned)
    from voxblox_amd import multi_gpu, scenes
    for rank, _, _, _, owned in results:
        keys = np.array(list(owned.keys()), np.int32).reshape(-1, 3)
        if keys.shape[0]:
            assert np.all(multi_gpu.owner_of(keys, world) == rank)
    # serial restatement of all five frames (warm-up frames land in the same persistent map): per frame, every rank's cloud
    # into a fresh delta, merged in rank order
    voxel = 0.1
    cfg = oracle.tsdf_cfg(default_truncation_distance=4 * voxel, integrator_threads=1)
    ref = {}
    for k in range(5):
        step = {}   # the owner adds the rows of one block in sender order, then folds the sum in ONCE (vbx_blocks_merge_sums)
        for rank in range(world):
            pose, pts, col = scenes.room_frame((k + rank * 100 // world) % 100, 100, f=40.0, width=80, height=60)
            oracle.lib().orc_fast_reset_counter_set(0)
            m = oracle.OracleMap(voxel, 16)
            m.tsdf_integrator("fast", cfg).integrate(pose[0], pose[1], pts, col)
            for key, (d, w, c, _) in m.tsdf_dict().items():
                sA = np.stack([w * d, w] + [w * c[:, ch].astype(np.float32) for ch in range(4)])
                step[key] = step[key] + sA if key in step else sA
        for key, sA in step.items():
            if not (sA[1] > 0).any():
                continue
            dB, wB, cB = ref.get(key, (np.zeros(4096, np.float32), np.zeros(4096, np.float32), np.zeros((4096, 4), np.uint8)))
            ref[key] = merge_A_into_B(sA, dB, wB, cB)
    assert set(merged) == set(ref), (len(merged), len(ref))
    for key in ref:
        gd, gw, gc, _ = merged[key]
        rd, rw, rc = ref[key]
        assert np.array_equal(gw > 0, rw > 0)
        assert np.allclose(gw, rw, rtol=1e-5, atol=1e-6)
        assert np.abs(gd - rd).max() <= 1e-5
        assert np.abs(gc.astype(np.int32) - rc.astype(np.int32)).max() <= 1
    assert len({multi_gpu.owner_of(np.array([k], np.int32), world)[0] for k in merged}) == world   # every rank owns something
