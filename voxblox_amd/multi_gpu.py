"""Ray-bundle sharding across the GPUs of one node with a sparse block merge over RCCL.

SURVEY.md §8(e).  One process per GPU.  Per frame every rank integrates ITS shards of the
rays (whole sensors, or contiguous bands of a cloud) into a zero-initialised per-frame delta
map; overlapping block updates are then combined with the reference's own merge semantics —
Block::mergeBlock / mergeVoxelAIntoVoxelB (core/block_inl.h:112-129,
src/utils/voxel_utils.cc:10-22) is a weighted sum, so every touched block of the delta map travels
to the block's OWNER (owner = hash(BlockIndex) mod world), which forms the partial sums (w*d, w, w*r, w*g,
w*b, w*a), adds them up and folds them into its shard of the persistent map:

  1. every rank lists the blocks its delta touched, grouped by owner            (host, tiny)
  2. all-to-all of the group sizes, then of the BlockIndex rows                 (world ints, 12 B / block)
  3. vbx_blocks_export_sums writes the touched blocks' rows in the same order — three planes of 32-bit words:
     distance, weight, colour, i.e. the delta voxels themselves, 48 KiB per block at vps 16 (rounds 1-4 sent the
     six products, 96 KiB; the owner now forms them with the same float operations, so the merged map is bit for
     bit the same) — and one all-to-all-v moves each group to its owner: ONLY touched blocks travel
  4. the owner sums the rows of equal BlockIndex in (sender rank, key) order and merges the
     result into the stored voxels                                              (vbx_blocks_merge_sums)

Round 1 staged a dense zero-padded [world * L, 6, vps^3] buffer on every rank and
reduce-scattered all of it (~100 MB per rank and frame at 0.05 m although most blocks are touched
by one or two ranks); the sparse exchange sends each rank's touched blocks once.

The persistent map is distributed by block ownership; no rank holds all of it.  Shard-then-merge
is not bit-identical to integrating the whole cloud into one map (the clamp of updateTsdfVoxel is
applied per delta): parity is defined against the same shard + merge done with the
CPU oracle (tests/test_multi_gpu_gloo.py).

The collective layer is torch.distributed ("nccl" is RCCL on ROCm; "gloo" for the CPU tests).
"""
import time

import numpy as np


import os as _os
_FULL_CLEAR = bool(_os.environ.get("VBX_DELTA_FULL_CLEAR"))   # A/B switch: delta maps released completely every step
_FORCE_KEEP = _os.environ.get("VBX_DELTA_KEEP") == "1"   # measurement switch: every delta map keeps its slots


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


def group_by_owner(keys, world):
    """keys (n,3) -> (keys reordered: owner 0's rows first, each group in (z,y,x) order; counts[world])."""
    k = _sort_rows_zyx(keys)
    own = owner_of(k, world)
    order = np.argsort(own, kind="stable")
    counts = np.bincount(own, minlength=int(world)).astype(np.int64)
    return np.ascontiguousarray(k[order]), counts


ROW_PLANES = 3        # a row of the exchange = the delta block itself: distance, weight, colour planes (48 KiB at vps 16)
BANDS_PER_SENSOR = 4   # ray bundles per 640x480 frame (76,800 rays each)


def deal_sensor_units(world, n_sensors=4, bands=None):
    """BASELINE configs[4]'s ray-bundle shards: n_sensors x B contiguous bands of a frame, dealt out in order
    (sensor-major).  Returns units[rank] = [(sensor, band, bands)].

    B is a property of the sharded integrator, not of the number of GPUs: BANDS_PER_SENSOR = 4 for every world up to
    16 ranks (more bands only when there are more ranks than shards), so the merged map — which depends on the shard
    layout and on nothing else — is the same map on 1, 2, 4, 8 and 16 GPUs.  A quarter frame per bundle is the measured
    optimum on one MI355X (68.9 / 53.3 / 42.7-46.2 / 57.5 ms per four-sensor step with 1 / 2 / 4 / 8 bands through the
    Python host path, 38.0 with 4 through libvbx_shard.so): the observed-set replay of a bundle is a chain of dependent
    rounds whose length grows with the rays that interact (22 rounds for a quarter frame, 315 for a whole one), and the
    bundles of a step run concurrently.  `bands` overrides it (the parity tests also run whole sensors and halves)."""
    world = max(int(world), 1)
    if bands is None:
        bands = max(int(_os.environ.get("VBX_SHARD_BANDS", BANDS_PER_SENSOR)), (world + n_sensors - 1) // n_sensors)
    units = [(s, b, bands) for s in range(n_sensors) for b in range(bands)]
    per = (len(units) + world - 1) // world
    return [units[r * per:(r + 1) * per] for r in range(world)]


def band_of(n_points, band, bands):
    """[lo, hi) of a cloud's `band`-th of `bands` contiguous ray bands (row-major clouds: image stripes)."""
    return band * n_points // bands, (band + 1) * n_points // bands


class ShardedTsdfMap:
    """persistent / delta(s): objects with the small backend protocol used below (voxblox_amd.multi_gpu.GpuBackend
    for the HIP path).  `delta` may be ONE delta map or a LIST of them: shard i of a step goes into delta
    i % len(deltas) — with as many deltas as shards every ray shard (sensor, or band of a sensor) has a delta map of its
    own, the shards of a step are integrated CONCURRENTLY (one host thread and one HIP stream per delta map; the
    integration is a chain of small dependent launches that leaves most of the chip idle) and the merged map depends
    on the shard layout only, not on how the shards were dealt to the ranks.  Shards that share a delta map are
    integrated one after the other into it (one delta for everything = the sequential form)."""

    def __init__(self, persistent, delta, rank, world, dist=None, apply_caps=False,
                 truncation=0.0, max_weight=0.0):
        self.p = persistent
        self.deltas = list(delta) if isinstance(delta, (list, tuple)) else [delta]
        self.d = self.deltas[0]
        for d in self.deltas:   # a delta map is rebuilt every step and nobody asks for its Layer order: no first-touch ranks
            off = getattr(getattr(d, "m", None), "set_block_order_tracking", None)
            if off is not None:
                off(False)
        self.rank, self.world = int(rank), int(world)
        self.dist = dist
        self.apply_caps, self.trunc, self.max_weight = apply_caps, truncation, max_weight
        self.last = {}
        self._used = 1
        # run the collectives even with one rank (exercises the RCCL calls on a 1-GPU box)
        import os
        self.force_collectives = bool(os.environ.get("VBX_FORCE_COLLECTIVES")) and dist is not None

    def _collective(self):
        return self.world > 1 or self.force_collectives

    def _comm_device(self):
        import torch
        # RCCL moves device tensors; gloo (CPU tests) moves host tensors
        return self.d.device if self.dist.get_backend() == "nccl" else torch.device("cpu")

    # -- one frame --------------------------------------------------------------------------
    def integrate_shard(self, kind, cfg, pos, quat, points, colors, n_points=None):
        """Integrate one shard of rays into a fresh delta map, send every touched block to its
        owner, fold the blocks this rank owns into its persistent shard."""
        self.integrate_shards(kind, cfg, [(pos, quat, points, colors, n_points)])

    def integrate_only(self, kind, cfg, shards):
        """The integration half of a step: shard i into delta i % len(deltas), the deltas concurrently."""
        nd = len(self.deltas)
        self._used = max(1, min(nd, len(shards)))
        groups = [[sh for i, sh in enumerate(shards) if i % nd == u] for u in range(self._used)]

        def run(u):
            d = self.deltas[u]
            d.clear()
            for pos, quat, points, colors, n in groups[u]:
                d.integrate(kind, cfg, pos, quat, points, colors, n)

        if not shards:
            self.deltas[0].clear()
        elif self._used == 1:
            run(0)
        else:
            import threading
            errs = []

            def guarded(u):
                try:
                    run(u)
                except BaseException as e:   # re-raised on the caller's thread
                    errs.append(e)

            th = [threading.Thread(target=guarded, args=(u,), name=f"vbx-shard-{u}") for u in range(1, self._used)]
            for t in th:
                t.start()
            guarded(0)
            for t in th:
                t.join()
            if errs:
                raise errs[0]

    def integrate_shards(self, kind, cfg, shards):
        """The shards of one time step on this rank (e.g. two sensors at world 2), then ONE exchange.
        shards: [(pos, quat, points, colors, n)]."""
        self.integrate_only(kind, cfg, shards)
        self.exchange_and_merge()

    def exchange_and_merge(self):
        import torch
        nvox = self.d.nvox
        W = self.world
        # rows per (owner, delta): owner-major, then delta order, then (z,y,x) — at the owner the rows of one block
        # therefore arrive in (sender rank, delta, key) order = the global shard order, whatever the world size
        per = [group_by_owner(d.block_indices(), W) for d in self.deltas[:self._used]]
        send_counts = np.zeros(W, np.int64)
        for _, c in per:
            send_counts += c
        n_send = int(send_counts.sum())
        if len(per) == 1:
            send_keys = per[0][0]
            send = self.d.zeros((max(n_send, 1), ROW_PLANES, nvox))
            if n_send:
                self.d.export_sums(send_keys, send[:n_send])
        else:
            tmp = []
            for (k, c), d in zip(per, self.deltas):
                t = d.zeros((max(int(k.shape[0]), 1), ROW_PLANES, nvox))
                if k.shape[0]:
                    d.export_sums(k, t[:k.shape[0]])
                tmp.append(t)
            key_parts, row_parts = [], []
            offs = [np.concatenate([[0], np.cumsum(c)]) for _, c in per]
            for o in range(W):
                for u, (k, c) in enumerate(per):
                    a, b = int(offs[u][o]), int(offs[u][o + 1])
                    if b > a:
                        key_parts.append(k[a:b])
                        row_parts.append(tmp[u][a:b])
            send_keys = np.ascontiguousarray(np.concatenate(key_parts)) if key_parts else np.zeros((0, 3), np.int32)
            send = torch.cat(row_parts) if row_parts else self.d.zeros((1, ROW_PLANES, nvox))
        if not self._collective():
            recv_keys, recv = send_keys, send[:n_send]
            recv_counts = send_counts
        else:
            dev = self._comm_device()
            sc = torch.from_numpy(send_counts).to(dev)
            rc = torch.zeros_like(sc)
            self.dist.all_to_all_single(rc, sc)
            recv_counts = rc.cpu().numpy()
            n_recv = int(recv_counts.sum())
            kt = torch.from_numpy(send_keys.reshape(-1)).to(dev)
            rk = torch.empty(n_recv * 3, dtype=torch.int32, device=dev)
            self.dist.all_to_all_single(rk, kt, output_split_sizes=[int(c) * 3 for c in recv_counts],
                                        input_split_sizes=[int(c) * 3 for c in send_counts])
            recv_keys = rk.cpu().numpy().reshape(-1, 3)
            payload = send[:n_send].reshape(n_send, ROW_PLANES * nvox)
            if payload.device != dev:
                payload = payload.to(dev)
            recv = torch.empty((n_recv, ROW_PLANES * nvox), dtype=torch.float32, device=dev)
            self.dist.all_to_all_single(recv, payload, output_split_sizes=[int(c) for c in recv_counts],
                                        input_split_sizes=[int(c) for c in send_counts])
            recv = recv.reshape(n_recv, ROW_PLANES, nvox)
            if recv.device != self.p.device:
                recv = recv.to(self.p.device)
        if recv_keys.shape[0]:
            # rows arrive grouped by sender rank, each group in (delta, z,y,x) order: the owner adds the rows
            # of one block in exactly that order (deterministic), then merges once
            self.p.merge_sums(recv_keys, recv, self.apply_caps, self.trunc, self.max_weight)
        self.last = dict(sent_blocks=n_send, received_blocks=int(recv_keys.shape[0]),
                         owned_blocks=int(np.unique(recv_keys, axis=0).shape[0]) if recv_keys.shape[0] else 0,
                         payload_bytes=int(n_send * ROW_PLANES * nvox * 4),
                         kept_local_bytes=int(send_counts[self.rank] * ROW_PLANES * nvox * 4) if self.rank < len(send_counts) else 0)


class PipelinedShardedTsdfMap:
    """The same frame step with the exchange pipelined behind the next frame's integration.

    Two delta maps (or two sets of per-shard delta maps, see ShardedTsdfMap) alternate: while a worker thread
    runs frame k's exchange (export, RCCL all-to-all, owner merge) on its own HIP stream, the caller already
    integrates frame k+1 into the other set.  The integration is latency-bound (the GPU is mostly idle between its
    kernels), so the two overlap well; per-frame time tends to max(integrate, exchange) instead of
    their sum.  Every collective is issued by the worker thread, in frame order, so all ranks issue
    them in the same order; call flush() before any collective of your own (barriers) and before
    reading the persistent map."""

    def __init__(self, persistent, deltas, rank, world, dist=None, apply_caps=False, truncation=0.0,
                 max_weight=0.0, device=None):
        import queue
        import sys
        import threading
        assert len(deltas) == 2   # two delta maps, or two equally long LISTS of delta maps (one per concurrent shard)
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
        self.stats = {"frames": 0, "wait_s": 0.0, "integrate_s": 0.0, "exchange_s": 0.0, "payload_bytes": 0,
                      "sent_blocks": 0}
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
                    self.stats["payload_bytes"] += self.last["payload_bytes"]
                    self.stats["sent_blocks"] += self.last["sent_blocks"]
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

    def integrate_shards(self, kind, cfg, shards):
        i = self._n & 1
        self._n += 1
        t0 = time.perf_counter()
        self._idle[i].wait()          # this delta map's previous exchange is done
        t1 = time.perf_counter()
        self._check()
        self._idle[i].clear()
        self.sm[i].integrate_only(kind, cfg, shards)
        self._q.put(i)
        self.stats["wait_s"] += t1 - t0
        self.stats["integrate_s"] += time.perf_counter() - t1
        self.stats["frames"] += 1

    def integrate_shard(self, kind, cfg, pos, quat, points, colors, n_points=None):
        self.integrate_shards(kind, cfg, [(pos, quat, points, colors, n_points)])

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

    def __init__(self, gmap, device, keep_slots=False):
        """keep_slots: a delta map that sees the same region step after step keeps its blocks' pool slots when it is
        cleared (vbx_clear_keep_slots) instead of allocating them again every step.  Measured (MI355X): 68.4 -> 67.0 ms per
        four-sensor step at 0.02 m, but 1.78 -> 2.05 ms per frame for one 0.05 m sensor with the exchange pipelined
        behind it — so bench.py switches it on for configs[4] only."""
        import torch
        self.keep_slots = bool(keep_slots)
        self.m = gmap
        self.device = torch.device(device)
        self.nvox = gmap.vps ** 3
        self._torch = torch

    def clear(self):
        # a delta map sees the same region step after step: its blocks leave the layer but keep their pool slots
        if (self.keep_slots or _FORCE_KEEP) and not _FULL_CLEAR:
            self.m.clear_keep_slots()
        else:
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
        # export_sums writes every plane of every listed block: no need to clear the staging
        return self._torch.empty(shape, dtype=self._torch.float32, device=self.device)

    def export_sums(self, keys, out_view):
        assert out_view.is_contiguous()
        # vbx_blocks_export_sums synchronises its own stream before returning; the torch stream
        # that reads the buffer next is ordered behind this host call
        self.m.export_sums(keys, out_view.data_ptr())

    def merge_sums(self, keys, sums, apply_caps, trunc, max_weight):
        sums = sums.contiguous()
        self._torch.cuda.current_stream(self.device).synchronize()   # the all-to-all that filled `sums`
        self.m.merge_sums(keys, sums.data_ptr(), apply_caps, trunc, max_weight)
