"""Ray-bundle sharding across the GPUs of one node with a block merge over RCCL.

SURVEY.md §8(e).  One process per GPU.  Per frame every rank integrates ITS shard of the
rays (a whole sensor, or a contiguous bundle of a cloud) into a zero-initialised per-frame
delta map; overlapping block updates are then combined with the reference's own merge
semantics — Block::mergeBlock / mergeVoxelAIntoVoxelB (core/block_inl.h:112-129,
src/utils/voxel_utils.cc:10-22) is a weighted sum, hence a collective:

  1. all-gather of each rank's touched BlockIndex list  ->  every rank builds the same union,
     grouped by owner rank (owner = hash(BlockIndex) mod world), each group padded to the
     same length L so that chunk r of the staging buffer is exactly rank r's blocks;
  2. every rank writes its delta voxels as partial sums (w*d, w, w*r, w*g, w*b, w*a) into a
     dense [world*L, 6, vps^3] buffer (zeros where it has nothing)   (vbx_blocks_export_sums);
  3. reduce-scatter(sum): rank r receives the summed deltas of the blocks it owns (RCCL over
     xGMI; payload 96 KiB per union block at vps 16);
  4. the owner folds them into its shard of the persistent map        (vbx_blocks_merge_sums).

The persistent map is therefore distributed by block ownership; no rank holds all of it.
Shard-then-merge is not bit-identical to integrating the whole cloud into one map (the clamp
of updateTsdfVoxel is applied per delta): parity is defined against the same
shard + merge done with the CPU oracle (tests/test_multi_gpu_gloo.py).

The collective layer is torch.distributed ("nccl" is RCCL on ROCm; "gloo" for the CPU tests).
"""
import time

import numpy as np

_SENTINEL = np.iinfo(np.int32).max


def owner_of(keys, world):
    """Deterministic owner rank of each BlockIndex row (n,3) -> (n,) int64."""
    k = np.asarray(keys, np.int64).reshape(-1, 3)
    h = (k[:, 0] * 73856093) ^ (k[:, 1] * 19349663) ^ (k[:, 2] * 83492791)
    return (h & 0x7FFFFFFF) % max(int(world), 1)


def _sort_rows_zyx(keys):
    k = np.asarray(keys, np.int32).reshape(-1, 3)
    if k.shape[0] == 0:
        return k
    order = np.lexsort((k[:, 0], k[:, 1], k[:, 2]))
    return k[order]


def build_layout(all_keys, world):
    """all_keys: list (one per rank) of (n_r,3) int32 arrays.  Returns (union_by_owner, L):
    union_by_owner[r] = sorted unique keys owned by rank r; L = padded group length."""
    cat = np.concatenate([np.asarray(k, np.int32).reshape(-1, 3) for k in all_keys], 0) \
        if all_keys else np.zeros((0, 3), np.int32)
    uni = np.unique(cat, axis=0) if cat.shape[0] else cat
    uni = _sort_rows_zyx(uni)
    own = owner_of(uni, world)
    groups = [uni[own == r] for r in range(world)]
    L = max([g.shape[0] for g in groups] + [1])
    return groups, L


class ShardedTsdfMap:
    """persistent / delta: objects with the small backend protocol used below
    (voxblox_amd.multi_gpu.GpuBackend for the HIP path)."""

    def __init__(self, persistent, delta, rank, world, dist=None, apply_caps=False,
                 truncation=0.0, max_weight=0.0):
        self.p, self.d = persistent, delta
        self.rank, self.world = int(rank), int(world)
        self.dist = dist
        self.apply_caps, self.trunc, self.max_weight = apply_caps, truncation, max_weight
        self.last = {}
        # run the collectives even with one rank (exercises the RCCL calls on a 1-GPU box)
        import os
        self.force_collectives = bool(os.environ.get("VBX_FORCE_COLLECTIVES")) and dist is not None

    # -- collectives ------------------------------------------------------------------------
    def _gather_keys(self, keys):
        import torch
        if self.world == 1 and not self.force_collectives:
            return [keys]
        # RCCL moves device tensors; gloo (CPU tests) gathers host tensors
        dev = self.d.device if self.dist.get_backend() == "nccl" else torch.device("cpu")
        n = torch.tensor([keys.shape[0]], dtype=torch.int64, device=dev)
        counts = [torch.zeros_like(n) for _ in range(self.world)]
        self.dist.all_gather(counts, n)
        counts = [int(c.item()) for c in counts]
        m = max(max(counts), 1)
        pad = np.full((m, 3), _SENTINEL, np.int32)
        pad[:keys.shape[0]] = keys
        mine = torch.from_numpy(pad).to(dev)
        out = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        return [o.cpu().numpy()[:c] for o, c in zip(out, counts)]

    def _reduce_scatter(self, sums, L):
        """sums: [world*L, 6, nvox] -> this rank's [L, 6, nvox] chunk of the elementwise sum."""
        import torch
        if self.world == 1 and not self.force_collectives:
            return sums
        if self.dist.get_backend() == "nccl":
            out = torch.empty((L,) + tuple(sums.shape[1:]), dtype=sums.dtype, device=sums.device)
            self.dist.reduce_scatter_tensor(out, sums, op=self.dist.ReduceOp.SUM)
            return out
        self.dist.all_reduce(sums, op=self.dist.ReduceOp.SUM)  # gloo has no reduce-scatter
        return sums[self.rank * L:(self.rank + 1) * L]

    # -- one frame --------------------------------------------------------------------------
    def integrate_shard(self, kind, cfg, pos, quat, points, colors, n_points=None):
        """Integrate this rank's rays into the delta map, merge all ranks' deltas, fold the
        blocks this rank owns into its persistent shard."""
        self.d.clear()
        self.d.integrate(kind, cfg, pos, quat, points, colors, n_points)
        self.exchange_and_merge()

    def exchange_and_merge(self):
        keys = _sort_rows_zyx(self.d.block_indices())
        groups, L = build_layout(self._gather_keys(keys), self.world)
        nvox = self.d.nvox
        sums = self.d.zeros((self.world * L, 6, nvox))
        for r, g in enumerate(groups):
            if g.shape[0]:
                self.d.export_sums(g, sums[r * L:r * L + g.shape[0]])
        mine = self._reduce_scatter(sums, L)
        g = groups[self.rank]
        if g.shape[0]:
            self.p.merge_sums(g, mine[:g.shape[0]], self.apply_caps, self.trunc, self.max_weight)
        self.last = dict(union_blocks=int(sum(x.shape[0] for x in groups)), owned_blocks=int(g.shape[0]),
                         padded_rows=int(self.world * L), payload_bytes=int(self.world * L * 6 * nvox * 4))


class PipelinedShardedTsdfMap:
    """The same frame step with the exchange pipelined behind the next frame's integration.

    Two delta maps alternate: while a worker thread runs frame k's exchange (key all-gather,
    export, RCCL reduce-scatter, owner merge) on its own HIP stream, the caller already
    integrates frame k+1 into the other delta map.  The integration is latency-bound (the GPU is
    mostly idle between its kernels), so the two overlap well; per-frame time tends to
    max(integrate, exchange) instead of their sum.  Every collective is issued by the worker
    thread, in frame order, so all ranks issue them in the same order; call flush() before any
    collective of your own (barriers) and before reading the persistent map."""

    def __init__(self, persistent, deltas, rank, world, dist=None, apply_caps=False, truncation=0.0,
                 max_weight=0.0, device=None):
        import queue
        import sys
        import threading
        assert len(deltas) == 2
        # the caller's thread comes back from a ~1 ms native call and must not wait long for the
        # GIL while the worker is between its own native calls (default switch interval: 5 ms)
        sys.setswitchinterval(5e-5)
        self.sm = [ShardedTsdfMap(persistent, d, rank, world, dist, apply_caps, truncation, max_weight)
                   for d in deltas]
        self.p = persistent
        self.device = device
        self._q = queue.Queue()
        self._idle = [threading.Event(), threading.Event()]
        for e in self._idle:
            e.set()
        self._err = None
        self._n = 0
        self.last = {}
        self.stats = {"frames": 0, "wait_s": 0.0, "integrate_s": 0.0, "exchange_s": 0.0}
        self._thread = threading.Thread(target=self._worker, name="vbx-exchange", daemon=True)
        self._thread.start()

    def _worker(self):
        stream_ctx = None
        if self.device is not None:
            import torch
            torch.cuda.set_device(self.device)
            stream_ctx = torch.cuda.stream(torch.cuda.Stream(self.device))
            stream_ctx.__enter__()
        try:
            while True:
                i = self._q.get()
                if i is None:
                    return
                try:
                    t0 = time.perf_counter()
                    self.sm[i].exchange_and_merge()
                    self.stats["exchange_s"] += time.perf_counter() - t0
                    self.last = self.sm[i].last
                except BaseException as e:  # surfaced by the next call on the caller's thread
                    self._err = e
                self._idle[i].set()
        finally:
            if stream_ctx is not None:
                stream_ctx.__exit__(None, None, None)

    def _check(self):
        if self._err is not None:
            e, self._err = self._err, None
            raise e

    def integrate_shard(self, kind, cfg, pos, quat, points, colors, n_points=None):
        i = self._n & 1
        self._n += 1
        t0 = time.perf_counter()
        self._idle[i].wait()          # this delta map's previous exchange is done
        t1 = time.perf_counter()
        self._check()
        self._idle[i].clear()
        d = self.sm[i].d
        d.clear()
        d.integrate(kind, cfg, pos, quat, points, colors, n_points)
        self._q.put(i)
        self.stats["wait_s"] += t1 - t0
        self.stats["integrate_s"] += time.perf_counter() - t1
        self.stats["frames"] += 1

    def flush(self):
        for e in self._idle:
            e.wait()
        self._check()

    def close(self):
        self.flush()
        self._q.put(None)
        self._thread.join(30)


class GpuBackend:
    """HIP map (voxblox_amd.capi.Map) + torch device tensors for the staging buffers."""

    def __init__(self, gmap, device):
        import torch
        self.m = gmap
        self.device = torch.device(device)
        self.nvox = gmap.vps ** 3
        self._torch = torch

    def clear(self):
        self.m.clear()

    def integrate(self, kind, cfg, pos, quat, points, colors, n_points=None):
        if hasattr(points, "data_ptr"):
            self.m.integrate_device(kind, cfg, pos, quat, points.data_ptr(), colors.data_ptr(),
                                    int(n_points if n_points is not None else points.shape[0]))
        else:
            self.m.integrate(kind, cfg, pos, quat, points, colors)

    def block_indices(self):
        return self.m.block_indices()

    def zeros(self, shape):
        return self._torch.zeros(shape, dtype=self._torch.float32, device=self.device)

    def export_sums(self, keys, out_view):
        assert out_view.is_contiguous()
        self._torch.cuda.current_stream(self.device).synchronize()
        self.m.export_sums(keys, out_view.data_ptr())

    def merge_sums(self, keys, sums, apply_caps, trunc, max_weight):
        sums = sums.contiguous()
        self._torch.cuda.current_stream(self.device).synchronize()
        self.m.merge_sums(keys, sums.data_ptr(), apply_caps, trunc, max_weight)
