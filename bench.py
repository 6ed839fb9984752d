#!/usr/bin/env python
"""bench.py — Mpoints/s integrated by the HIP TSDF hot path on MI355X (BASELINE.json's metric).

Workloads
  stream    BASELINE configs[1] (default at --gpus 1): FastTsdfIntegrator, 640x480 synthetic room-scan
            stream (voxblox_amd.scenes.room_frame), 0.05 m voxels / 16^3 blocks, truncation 4 voxels, every
            other Config default.  One step = one integratePointCloud() call on one 307,200-point frame
            whose points/colours are already resident in HBM (vbx_tsdf_integrate_device).
            Variants: --integrator merged --scene cow (configs[2]), --esdf (configs[3]), --mesh, --voxel.
  stream at --gpus N > 1 (the default there too): the same configs[1] stream on EVERY GPU — one sensor per rank, a quarter
            turn apart in the same room — each rank integrating its cloud into a per-frame delta map, the touched blocks'
            weighted sums going to their owner ranks over RCCL (sparse all-to-all-v, pipelined behind the next frame) and
            merged there into ONE map distributed by block owner: weak scaling of the metric's own workload
            (`value` = N x 307,200 points per step / time); BASELINE configs[4] runs as a short secondary leg of the same launch.
  sensors4  BASELINE configs[4] (`--workload sensors4`): four concurrent 640x480 sensors, 0.02 m voxels,
            sixteen ray bundles per step (four contiguous bands per frame — the same layout for every N, so the merged
            map does not depend on the number of GPUs) dealt over the ranks, every rank integrating its bundles
            concurrently into per-step delta maps of their own, a sparse RCCL all-to-all
            of the touched blocks' weighted sums to the block owners, owner merge into the persistent map
            (voxblox_amd/multi_gpu.py).  One step = all four sensors' frames; the total work
            per step is the same for every N (strong scaling), and `--gpus 1 --workload sensors4` runs
            the very same shard + merge on one GPU.

Contract: W untimed warm-up steps, then exactly K timed steps bracketed by barrier +
torch.cuda.synchronize() on both sides, MAX over ranks, rank 0 prints ONE JSON line.
`--gpus N` without a launcher spawns the N ranks itself (torch.distributed.run, 127.0.0.1) and fails loudly
when the box has fewer GPUs; under a launcher (RANK set) it reads RANK / LOCAL_RANK / WORLD_SIZE.

The JSON line's `roofline` block is computed from the per-kernel table measured in this run (vbx_profile_*:
two HIP events around every launch on the launch stream, a separate pass over the same stream right after
the timed region so that the events stay out of `value`); `cpu_baseline` times the reference's own sources
(oracle/_ref) on this host for integrator_threads in {1,2,4,...,nproc}.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

VOXEL = 0.05
N_STREAM = 100
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default 60 for stream, 25 for sensors4)")
    ap.add_argument("--warmup", type=int, default=-1, help="untimed steps (default 5 for stream, 2 for sensors4)")
    ap.add_argument("--workload", default="auto", choices=["auto", "stream", "sensors4"])
    ap.add_argument("--integrator", default="fast", choices=["fast", "merged", "simple"])
    ap.add_argument("--esdf", action="store_true",
                    help="BASELINE configs[3]: EsdfIntegrator::updateFromTsdfLayer(true) after every frame")
    ap.add_argument("--mesh", action="store_true",
                    help="MeshIntegrator::generateMesh(true, true) after every frame (SURVEY 8(f) #4)")
    ap.add_argument("--scene", default="room", choices=["room", "cow"],
                    help="room = configs[1]/[3] stream; cow = configs[2] Cow-and-Lady-style orbit")
    ap.add_argument("--voxel", type=float, default=0.0, help="voxel size (default 0.05 for stream, 0.02 for sensors4)")
    ap.add_argument("--max-blocks", type=int, default=0, help="block pool capacity (0 = sized from --voxel)")
    ap.add_argument("--merged-order", type=int, default=0, choices=[0, 1])
    ap.add_argument("--fast-set", type=int, default=0, choices=[0, 1])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the vbx_tsdf_integrate (host pointers) leg")
    ap.add_argument("--no-extras", action="store_true",
                    help="default run only: skip the short secondary legs (configs[2], configs[3], configs[4] on 1 GPU)")
    ap.add_argument("--mirror-frames", type=int, default=8, help="stream: frames that also mirror touched blocks to the host")
    ap.add_argument("--profile-frames", type=int, default=12, help="frames of the per-kernel profile pass (0 = skip)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "torch", "native"],
                    help="sensors4: torch = voxblox_amd.multi_gpu over torch.distributed (RCCL); native = libvbx_shard.so (the C++ host "
                         "path: its own threads per ray bundle, RCCL called directly); both run the exchange behind the next step.  "
                         "auto = native on one GPU (sixteen concurrent bundles: no interpreter lock between the launching threads), "
                         "torch on several (the path the multi-rank CPU tests cover)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU-reference sample")
    ap.add_argument("--esdf-fidelity-frames", type=int, default=24,
                    help="--esdf: frames of the lockstep GPU-vs-reference ESDF comparison (0 = skip)")
    ap.add_argument("--esdf-mode", default="reference", choices=["reference", "order_free"],
                    help="--esdf: what the timed region runs.  reference = vbx_esdf_cfg.reference_order = 1, the reference's own "
                         "result (checked voxel by voxel against the reference build after the clock stops); order_free = the "
                         "order-free fixed point (fast, NOT bit-exact)")
    ap.add_argument("--bands", type=int, default=0,
                    help="sensors4: ray bundles per sensor frame (default 1 = whole sensors, the only layout that reproduces the "
                         "reference's map; 4 = the quarter-frame bundles of rounds 3-4, a different map)")
    ap.add_argument("--cpu-threads", default="", help="comma list of integrator_threads values the CPU reference is timed with (default: 1, 2, 4, ... nproc)")
    ap.add_argument("--detail-out", default="", help="where the full result goes (default bench_detail.json next to this file)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# launcher: --gpus N spawns N ranks
# ------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn(args):
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs, this box has {have} "
                         "(one rank per GPU; RCCL refuses two ranks on one device)\n")
        sys.exit(2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


# ------------------------------------------------------------------------------------------------
# CPU baselines: the reference's own sources (oracle/_ref) when built, else the restatement
# ------------------------------------------------------------------------------------------------
def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    use_ref = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libvbxref.so"))
    return O, (O.ref_lib() if use_ref else O.lib()), use_ref


CPU_THREADS = []   # --cpu-threads (empty: the full ladder)


def host_cpu_info():
    """What the CPU figures were measured on: hardware threads the machine reports, the ones this process may run on, and the
    cgroup's CPU quota (a throttled lease would otherwise look like an idle 256-thread host)."""
    info = {"host_hw_threads": os.cpu_count() or 1}
    try:
        info["affinity_threads"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        info["affinity_threads"] = None
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota = None if txt[0] == "max" else round(float(txt[0]) / float(txt[1]), 2)
            else:
                q = float(txt[0])
                period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0])
                quota = None if q < 0 else round(q / period, 2)
            info["cgroup_cpu_quota_source"] = path
            break
        except (OSError, ValueError, IndexError):
            continue
    info["cgroup_cpu_quota_cores"] = quota   # None: unlimited (or no cgroup file readable)
    try:
        info["loadavg_1min"] = round(os.getloadavg()[0], 2)
    except OSError:
        pass
    return info


def _usable_cores():
    i = host_cpu_info()
    n = i.get("affinity_threads") or i["host_hw_threads"]
    if i.get("cgroup_cpu_quota_cores"):
        n = max(1, min(n, int(i["cgroup_cpu_quota_cores"])))
    return n


def _thread_counts(cores):
    if CPU_THREADS:
        return sorted({t for t in CPU_THREADS if 1 <= t <= cores}) or [1]
    return sorted({t for t in (1, 2, 4, 8, 16, 32, 64, cores) if t <= cores})


def cpu_baseline_frames(voxel):
    """Frames every thread count (and every leg of a run) is timed on: 2 warm-up + 10 timed at >= 0.05 m,
    2 + 4 at finer voxels (a 0.02 m frame costs the reference 0.5-1 s)."""
    return (2, 10) if voxel >= 0.049 else (2, 4)


def cpu_baseline(frames, kind, voxel, seconds=None, window=None):
    """integratePointCloud of the reference on a FIXED sample of the same stream — the same frame list for every
    thread count in {1,2,4,...,nproc} (its own spawn-per-call threading) and for every leg of a run, so that the
    figures are comparable.  window = (n_warm, n_timed): frames[0:n_warm] untimed (they build the map the timed
    frames land in), median over the next n_timed — the caller passes the GPU leg's own warm-up / timed split so
    that both sides are timed on the SAME frames; without it 2 + 10 (2 + 4 below 0.05 m).  `seconds` is ignored
    (kept for callers that still pass a budget): the sample is bounded by its frame count."""
    import ctypes
    O, L, use_ref = _oracle()
    cores = _usable_cores()
    counts = _thread_counts(cores)
    n_warm, n_timed = window or cpu_baseline_frames(voxel)
    sample = frames[:n_warm + n_timed]
    by = {}
    t_all = time.time()
    for threads in counts:
        L.orc_fast_reset_counter_set(0)
        m = O.OracleMap(voxel, 16, L=L)
        c = O.TsdfCfg()
        L.orc_tsdf_cfg_default(ctypes.byref(c))
        c.default_truncation_distance = 4 * voxel
        c.integrator_threads = threads
        it = m.tsdf_integrator(kind, c)
        ts = []
        for pose, pts, col in sample:
            t0 = time.perf_counter()
            it.integrate(pose[0], pose[1], pts, col)
            ts.append(time.perf_counter() - t0)
        used = ts[n_warm:] if len(ts) > n_warm else ts
        med = float(np.median(used))
        by[str(threads)] = {"value": round(sample[0][1].shape[0] / med / 1e6, 3), "median_ms": round(med * 1e3, 2),
                            "frames": len(ts)}
        del it, m
    best = max(by, key=lambda k: by[k]["value"])
    return {"value": by[best]["value"], "unit": "Mpoints/s", "cores": int(best),
            "kind": "reference" if use_ref else "port", "host_hw_threads": os.cpu_count() or 1, "host": host_cpu_info(), "by_threads": by,
            "cpu_seconds_spent": round(time.time() - t_all, 1),
            "frames_timed": [n_warm, len(sample) - 1],
            "sample": f"{kind} integrator, {voxel:g} m voxels, frames 0..{len(sample) - 1} of the same stream for EVERY thread count "
                      f"(the first {n_warm} untimed, median of the other {len(sample) - n_warm}"
                      + (" = the frames `value` is timed on" if window else "") + "), "
                      + ("reference sources compiled over dependency shims (oracle/_ref)" if use_ref
                         else "oracle restatement")}


def cpu_esdf_baseline(frames, voxel, seconds):
    """Reference Fast integration + EsdfIntegrator::updateFromTsdfLayer(true) after every frame
    (esdf_integrator.cc:104-122); the ESDF call is what is timed.  Single thread: the ESDF integrator has none."""
    import ctypes
    O, L, use_ref = _oracle()
    L.orc_fast_reset_counter_set(0)
    m = O.OracleMap(voxel, 16, L=L)
    c = O.TsdfCfg()
    L.orc_tsdf_cfg_default(ctypes.byref(c))
    c.default_truncation_distance = 4 * voxel
    c.integrator_threads = min(os.cpu_count() or 1, 8)
    it = m.tsdf_integrator("fast", c)
    ec = O.EsdfCfg()
    L.orc_esdf_cfg_default(ctypes.byref(ec))
    ec.min_distance_m = 2 * voxel
    e = m.esdf_integrator(ec)
    ts = []
    t_begin = time.time()
    for i, (pose, pts, col) in enumerate(frames):
        it.integrate(pose[0], pose[1], pts, col)
        t0 = time.perf_counter()
        e.update_from_tsdf_layer(True)
        ts.append(time.perf_counter() - t0)
        if time.time() - t_begin > seconds and i >= 4:
            break
    used = ts[2:] if len(ts) > 3 else ts
    st = e.stats()
    return {"ms_per_update": round(float(np.median(used)) * 1e3, 3), "cores": 1, "kind": "reference" if use_ref else "port",
            "frames": len(ts), "relaxations_per_update": round(st["relaxations"] / max(len(ts), 1), 1),
            "sample": "updateFromTsdfLayer(true) after every frame of the same stream, median after 2 warm-up frames"}


def esdf_fidelity(frames, voxel, n_frames, checkpoints):
    """configs[3] in lockstep on the HIP path and on the reference (oracle/_ref as the checker): Fast
    integration + incremental ESDF update after every frame; after every frame the GPU's ESDF layer is
    compared with the reference's incremental layer, and at the checkpoints with the reference's BATCH
    result of the same TSDF (updateFromTsdfLayerBatch on a second reference map).  Reported, not asserted:
    the device wavefront is order-free by design, the reference's result depends on its
    bucket-queue pop order and on min_diff_m; this is the measured distance between the two."""
    import ctypes
    from voxblox_amd import capi
    O, L, use_ref = _oracle()
    L.orc_fast_reset_counter_set(0)

    def ref_pair():
        m = O.OracleMap(voxel, 16, L=L)
        c = O.TsdfCfg()
        L.orc_tsdf_cfg_default(ctypes.byref(c))
        c.default_truncation_distance = 4 * voxel
        c.integrator_threads = 1
        ec = O.EsdfCfg()
        L.orc_esdf_cfg_default(ctypes.byref(ec))
        ec.min_distance_m = 2 * voxel
        return m, m.tsdf_integrator("fast", c), m.esdf_integrator(ec)

    mi, ti, ei = ref_pair()       # reference, incremental
    mb, tb, eb = ref_pair()       # reference, batch at the checkpoints
    gm = capi.Map(voxel, 16, max_blocks=8192)
    gs = capi.Map(voxel, 16, max_blocks=8192)   # the same stream with the ESDF in the reference's own order
    gcfg = capi.tsdf_cfg(default_truncation_distance=4 * voxel)
    ecfg = capi.esdf_cfg(min_distance_m=2 * voxel, reference_order=0)
    scfg = capi.esdf_cfg(min_distance_m=2 * voxel, reference_order=1)
    strict_ms = []

    def gpu_layer(m=None):
        m = m or gm
        idx = m.block_indices(capi.LAYER_ESDF)
        v, _, _ = m.blocks_download(idx, capi.LAYER_ESDF)
        return {tuple(int(x) for x in i): (v[k]["distance"], v[k]["observed"]) for k, i in enumerate(idx)}

    def compare(g, r):
        se = 0.0
        n = nd = n4 = 0
        worst = 0.0
        mask_diff = 0
        for k, (rd, rf, _, _) in r.items():
            obs = (rf & 1).astype(bool)
            if k not in g:
                mask_diff += int(obs.sum())
                continue
            gd, go = g[k]
            mask_diff += int((go.astype(bool) != obs).sum())
            both = obs & go.astype(bool)
            d = np.abs(gd[both] - rd[both])
            se += float((d.astype(np.float64) ** 2).sum())
            n += int(both.sum())
            nd += int((d > 0).sum())
            n4 += int((d > 1e-4).sum())
            if d.size:
                worst = max(worst, float(d.max()))
        return {"observed_voxels": n, "observed_mask_differences": mask_diff, "rmse_m": round((se / max(n, 1)) ** 0.5, 6),
                "max_m": round(worst, 5), "frac_differing": round(nd / max(n, 1), 5), "frac_gt_1e-4_m": round(n4 / max(n, 1), 5)}

    per_frame = []
    batch = {}
    for i, (pose, pts, col) in enumerate(frames[:n_frames]):
        gm.integrate(capi.TSDF_FAST, gcfg, pose[0], pose[1], pts, col)
        gm.esdf_update(ecfg, batch=False, clear_updated_flag=True)
        L.orc_fast_reset_counter_set(0)
        ti.integrate(pose[0], pose[1], pts, col)
        # reference order: the blocks carrying Update::kEsdf in the iteration order of the reference's own container
        # (Layer::getAllUpdatedBlocks, layer.h:194-203) are an input of the replay
        order = np.array([b for b in mi.block_indices(0) if mi.tsdf_block(b)[3] & 4], np.int32).reshape(-1, 3)
        ei.update_from_tsdf_layer(True)
        gs.integrate(capi.TSDF_FAST, gcfg, pose[0], pose[1], pts, col)
        t0 = time.perf_counter()
        gs.esdf_update_blocks(scfg, order, incremental=True)
        strict_ms.append((time.perf_counter() - t0) * 1e3)
        gs.clear_updated(capi.UPDATE_ESDF, capi.LAYER_TSDF)
        L.orc_fast_reset_counter_set(0)
        tb.integrate(pose[0], pose[1], pts, col)
        if (i + 1) in checkpoints or i + 1 == n_frames:
            g = gpu_layer()
            per_frame.append(dict(frame=i + 1, **compare(g, mi.esdf_dict())))
            eb.update_from_tsdf_layer_batch()
            batch[str(i + 1)] = compare(g, mb.esdf_dict())
    strict = compare(gpu_layer(gs), mi.esdf_dict())
    strict["ms_per_update"] = round(float(np.median(strict_ms)), 3)
    strict["note"] = ("vbx_esdf_cfg.reference_order = 1 (the reference's queue order replayed in parallel, DESIGN 4.5), same stream, block list "
                      "in the iteration order of the reference's own container, host wall clock of vbx_esdf_update_blocks incl. the list upload; "
                      "final layer against the reference's incremental layer: frac_differing must be 0")
    gm.close()
    gs.close()
    return {"frames": n_frames, "kind": "reference" if use_ref else "port",
            "reference_order_mode_vs_reference_incremental": strict,
            "vs_reference_incremental": per_frame, "vs_reference_batch": batch,
            "note": "GPU incremental stream (updateFromTsdfLayer(true) after every frame, default Config, min_diff_m 1e-3) "
                    "against the reference's own incremental stream and against its batch update of the same TSDF at the "
                    "checkpoints; distances in metres over voxels both sides observe"}


def esdf_reference_run(frames, voxel, n_frames):
    """configs[3] on the reference build (oracle/_ref, the checker): FastTsdfIntegrator with ONE thread (the deterministic
    reference the parity statement is about) + EsdfIntegrator::updateFromTsdfLayer(true) after every frame.  Returns the
    per-frame times of both calls, the block list every update walked (Layer::getAllUpdatedBlocks(Update::kEsdf), layer.h:194-203,
    in the container's own order) and the maps, so that the GPU leg can be compared voxel by voxel afterwards."""
    import ctypes
    O, L, use_ref = _oracle()
    L.orc_fast_reset_counter_set(0)
    m = O.OracleMap(voxel, 16, L=L)
    c = O.TsdfCfg()
    L.orc_tsdf_cfg_default(ctypes.byref(c))
    c.default_truncation_distance = 4 * voxel
    c.integrator_threads = 1
    it = m.tsdf_integrator("fast", c)
    ec = O.EsdfCfg()
    L.orc_esdf_cfg_default(ctypes.byref(ec))
    ec.min_distance_m = 2 * voxel
    e = m.esdf_integrator(ec)
    lists, t_tsdf, t_esdf = [], [], []
    for pose, pts, col in frames[:n_frames]:
        t0 = time.perf_counter()
        it.integrate(pose[0], pose[1], pts, col)
        t1 = time.perf_counter()
        lists.append(np.array([b for b in m.block_indices(0) if m.tsdf_block(b)[3] & 4], np.int32).reshape(-1, 3))
        t2 = time.perf_counter()
        e.update_from_tsdf_layer(True)
        t3 = time.perf_counter()
        t_tsdf.append((t1 - t0) * 1e3)
        t_esdf.append((t3 - t2) * 1e3)
    return {"map": m, "tsdf": it, "esdf": e, "lists": lists, "tsdf_ms": t_tsdf, "esdf_ms": t_esdf, "kind": "reference" if use_ref else "port",
            "all_blocks": m.block_indices(0)}


def esdf_layer_diff(gm, ref_esdf_dict):
    """Voxel-by-voxel comparison of the device's ESDF layer with the reference's: blocks, observed flags and distance BITS."""
    from voxblox_amd import capi
    idx = gm.block_indices(capi.LAYER_ESDF)
    v, _, _ = gm.blocks_download(idx, capi.LAYER_ESDF)
    g = {tuple(int(x) for x in i): k for k, i in enumerate(idx)}
    n = diff = worst_n = 0
    worst = 0.0
    se = 0.0
    n4 = 0
    for key, (rd, rf, _rp, _u) in ref_esdf_dict.items():
        obs = (rf & 1).astype(bool)
        n += int(obs.sum())
        if key not in g:
            diff += int(obs.sum())
            continue
        gd = v[g[key]]["distance"]
        go = v[g[key]]["observed"].astype(bool)
        bad = (go != obs) | (obs & (gd.view(np.uint32) != rd.view(np.uint32)))
        diff += int(bad.sum())
        both = obs & go
        d = np.abs(gd[both].astype(np.float64) - rd[both])
        if d.size:
            worst = max(worst, float(d.max()))
            se += float((d ** 2).sum())
            n4 += int((d > 1e-4).sum())
            worst_n += d.size
    extra = sum(int(v[k]["observed"].astype(bool).sum()) for key, k in g.items() if key not in ref_esdf_dict)
    return {"voxels_compared": n, "voxels_differing_from_reference": diff + extra, "max_abs_m": round(worst, 6),
            "rmse_m": round((se / max(worst_n, 1)) ** 0.5, 6), "frac_gt_1e-4_m": round(n4 / max(worst_n, 1), 5)}


DROPIN_EXTRA_BLOCKS = (5000, 20000)


def dropin_leg(frames, kind, voxel, warmup, steps, extra_blocks=0):
    import ctypes
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    L = O.ref_hip_lib()
    L.orc_fast_reset_counter_set(0)
    if hasattr(L, "orc_timing_reset"):
        L.orc_timing_reset()
    m = O.OracleMap(voxel, 16, L=L)
    if extra_blocks:
        # `extra_blocks` LOADED blocks far from the scene (io::LoadBlocksFromFile's effect on the host Layer: voxels written,
        # all Update bits set): they go to the device with the first call and are never touched again
        rng = np.random.RandomState(7)
        d = rng.uniform(-0.2, 0.2, 4096).astype(np.float32)
        w = rng.uniform(0.5, 50.0, 4096).astype(np.float32)
        col = rng.randint(0, 256, (4096, 4)).astype(np.uint8)
        side = int(np.ceil(extra_blocks ** (1.0 / 3.0)))
        k = 0
        for z in range(side):
            for y in range(side):
                for x in range(side):
                    if k < extra_blocks:
                        m.tsdf_block_set((1000 + x, 1000 + y, 1000 + z), d, w, col, 7)
                        k += 1
    c = O.TsdfCfg()
    L.orc_tsdf_cfg_default(ctypes.byref(c))
    c.default_truncation_distance = 4 * voxel
    it = m.tsdf_integrator(kind, c)
    ts = []
    tags = ("hip/tsdf_reconcile_from_host", "hip/tsdf_integrate_device", "hip/tsdf_mirror_to_host", "integrate/" + kind)
    before = {}
    for i in range(warmup + steps):
        pose, pts, col = frames[i % len(frames)]
        if i == warmup and hasattr(L, "orc_timing_get"):
            before = {t: O.timing_get(L, t) for t in tags}
        t0 = time.perf_counter()
        it.integrate(pose[0], pose[1], pts, col)
        ts.append(time.perf_counter() - t0)
    used = ts[warmup:]
    out = {"value": round(frames[0][1].shape[0] * len(used) / sum(used) / 1e6, 3), "unit": "Mpoints/s",
           "ms_per_step": round(sum(used) / len(used) * 1e3, 4), "frames": [warmup, warmup + steps - 1],
           "host_blocks_at_end": int(m.num_blocks(0)) if hasattr(m, "num_blocks") else None,
           "entry": "voxblox " + kind.capitalize() + "TsdfIntegrator::integratePointCloud (host Layer, host pointers) over the drop-in",
           "note": "wall time inside " + kind.capitalize() + "TsdfIntegrator::integratePointCloud of voxblox's own class over the C-ABI (SURVEY 8(d)'s "
                   "definition of the metric): reconcile of the host Layer, H2D of points + colours, device integration, touched blocks "
                   "mirrored back into the host Layer; same frames as `value`"}
    if before:
        split = {}
        for t in tags:
            n1, s1 = O.timing_get(L, t)
            n0, s0 = before[t]
            if n1 > n0:
                split[t] = round((s1 - s0) / (n1 - n0) * 1e3, 4)
        out["ms_by_timer_tag"] = split
        inside = split.get("integrate/" + kind)
        if inside:
            # the metric's own definition: the timing::Timer the reference itself puts around the body of integratePointCloud
            # (tsdf_integrator.cc:559 / :311 / :246); the perf_counter figure also holds the harness' conversion of the numpy
            # cloud into voxblox's Pointcloud / Colors containers, which a voxblox caller has already
            out["harness_wall_ms_per_step"] = out["ms_per_step"]
            out["ms_per_step"] = inside
            out["value"] = round(frames[0][1].shape[0] / inside / 1e3, 3)
    return out


def cpu_mesh_baseline(frames, kind, voxel):
    import ctypes
    O, L, use_ref = _oracle()
    cores = os.cpu_count() or 1
    res = {}
    for threads in (sorted({1, cores}) if use_ref else [1]):
        L.orc_fast_reset_counter_set(0)
        m = O.OracleMap(voxel, 16, L=L)
        c = O.TsdfCfg()
        L.orc_tsdf_cfg_default(ctypes.byref(c))
        c.default_truncation_distance = 4 * voxel
        c.integrator_threads = min(cores, 8)
        it = m.tsdf_integrator(kind, c)
        ml = m.mesh_layer()
        ts = []
        for pose, pts, col in frames:
            it.integrate(pose[0], pose[1], pts, col)
            t0 = time.perf_counter()
            ml.generate(True, True, threads=threads)
            ts.append(time.perf_counter() - t0)
        res[threads] = float(np.median(ts[2:]))
        del ml, it, m
    best = min(res, key=res.get)
    return {"ms_per_update": round(res[best] * 1e3, 3), "cores": best, "kind": "reference" if use_ref else "port",
            "sample": f"{len(frames)} frames, median after 2 warm-ups, best of integrator_threads in {sorted(res)}"}


# ------------------------------------------------------------------------------------------------
# per-kernel table -> roofline
# ------------------------------------------------------------------------------------------------
def kernel_table(gm, calls_per_step=1.0):
    raw, calls = gm.profile_table()
    tab = {}
    for k, (n, ms) in raw.items():      # template instances of one kernel count as that kernel (as rocprofv3 --stats
        k = k.split("<")[0]             # summaries do once aggregated by name)
        a = tab.setdefault(k, [0, 0.0])
        a[0] += n
        a[1] += ms
    steps = max(calls / calls_per_step, 1e-9)
    rows = [{"kernel": k, "launches_per_step": round(n / steps, 2), "avg_us": round(1e3 * ms / n, 2),
             "us_per_step": round(1e3 * ms / steps, 1)} for k, (n, ms) in tab.items() if n]
    rows.sort(key=lambda r: -r["us_per_step"])
    return rows, calls


def pmc_traffic(kernel, tag_files=("r06_pmc_hbm_traffic.json", "r05_pmc_hbm_traffic.json", "r03_pmc_hbm_traffic.json")):
    """HBM-side bytes per launch of `kernel` from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
    separate runs with --kernel-trace only, the driver-shaped command; tools/collect_profiles.sh +
    tools/summarize_profiles.py).  Counters cannot be read from inside this process, so the figure comes from the
    file; None when the file is missing or does not list the kernel."""
    for f in tag_files:
        path = os.path.join(ROOT, "profiles", f)
        if not os.path.exists(path):
            continue
        try:
            j = json.load(open(path))
            k = j["per_frame_bytes"].get(kernel)
            if not k or not k.get("launches_per_frame"):
                continue
            per_launch = (k["fetch_bytes"] + k["write_bytes"]) / k["launches_per_frame"]
            return {"bytes_per_step": int(k["fetch_bytes"] + k["write_bytes"]),
                    "bytes_per_launch": int(per_launch), "fetch_bytes_per_step": int(k["fetch_bytes"]),
                    "write_bytes_per_step": int(k["write_bytes"]), "launches_per_step": k["launches_per_frame"],
                    "all_kernels_bytes_per_step": int(j.get("total_bytes_per_frame", 0)),
                    "source": "profiles/" + f, "frames": j.get("frames_desc", j.get("frames")),
                    "corrections": j.get("units")}
        except Exception:
            continue
    return None


def roofline_from(rows, alg_bytes_per_step, device_ms_per_step, what):
    """The dominant kernel = the one with the largest time per step in the measured table.  `achieved` =
    the step's algorithmic bytes over that kernel's time per step (all of its launches in one step — the
    kernel is launched several times per frame and no single launch processes a frame's bytes); the same
    bytes over the whole step's device time is `step_frac`."""
    if not rows:
        return None
    dom = rows[0]
    t = dom["us_per_step"] * 1e-6
    achieved = alg_bytes_per_step / t / 1e9 if t > 0 else 0.0
    total_us = sum(r["us_per_step"] for r in rows)
    tr = (pmc_traffic(dom["kernel"]) if "distinct voxels updated per frame" in what else
          pmc_traffic(dom["kernel"], ("r06_pmc_esdf_ref_order.json", "r05_pmc_esdf_ref_order.json", "r03_pmc_esdf_traffic.json")) if "updated blocks" in what else None)
    return {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 6),
            # counter bytes PER STEP of the dominant kernel (all of its launches in a step), the same unit as
            # algorithmic_bytes_per_step and as `achieved`'s numerator: traffic / algorithmic_bytes_per_step is how many times
            # the compulsory bytes this kernel moves, traffic_all_kernels_per_step / algorithmic_bytes_per_step the whole step's
            "traffic": (tr["bytes_per_step"] if tr else None),
            "traffic_per_launch": (tr["bytes_per_launch"] if tr else None),
            "traffic_all_kernels_per_step": (tr["all_kernels_bytes_per_step"] if tr else None),
            "traffic_over_algorithmic": (round(tr["bytes_per_step"] / alg_bytes_per_step, 2) if tr and alg_bytes_per_step else None),
            "traffic_all_kernels_over_algorithmic": (round(tr["all_kernels_bytes_per_step"] / alg_bytes_per_step, 2)
                                                     if tr and alg_bytes_per_step and tr.get("all_kernels_bytes_per_step") else None),
            "traffic_detail": tr,
            "kernel": dom["kernel"], "launches_per_step": dom["launches_per_step"], "avg_launch_us": dom["avg_us"],
            "kernel_us_per_step": dom["us_per_step"], "algorithmic_bytes_per_step": int(alg_bytes_per_step),
            "algorithmic_bytes": what,
            "step_frac": round(alg_bytes_per_step / max(device_ms_per_step * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBPS, 6),
            "device_ms_per_step": round(device_ms_per_step, 4), "all_kernels_us_per_step": round(total_us, 1),
            "note": "durations from HIP events around every launch on the launch stream (a profiled pass over the "
                    "SAME frame indices as the timed region, on a second map brought to the same state by the same warm-up; events add ~2-4 us per launch, so short kernels read "
                    "high against rocprofv3 — profiles/ holds the matching rocprofv3 --kernel-trace --stats summary); "
                    "traffic = FETCH_SIZE + WRITE_SIZE bytes PER STEP of this kernel (all of its launches in a step; "
                    "traffic_per_launch beside it) from the committed PMC passes of the driver-shaped command "
                    "(traffic_detail.source; null when that file is absent), to be read against algorithmic_bytes_per_step; "
                    "latency / dependency bound path far below the HBM roofline (SURVEY 8(d))"}


# ------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------
def stream_frames(scene, rank, n, world=1):
    """The rank's sensor stream: the same trajectory, the ranks evenly spread along it (a quarter turn apart at 4 ranks)."""
    from voxblox_amd import scenes
    if scene == "cow":
        return [scenes.cow_and_lady_like_frame((k + (200 * rank) // max(world, 1)) % 200) for k in range(min(n, 200))]
    return [scenes.room_frame((k + (N_STREAM * rank) // max(world, 1)) % N_STREAM, N_STREAM) for k in range(min(n, N_STREAM))]


def to_device(frames, dev):
    import torch
    return [(pose, torch.from_numpy(pts).to(dev), torch.from_numpy(col).to(dev), pts.shape[0]) for pose, pts, col in frames]


def run_stream(args, gm, kind, cfg, d_frames, steps, warmup, barrier, esdf_cfg=None, mesh_cfg=None, esdf_lists=None):
    """W warm-up + K timed steps on one map; returns dt, per-stage ms, counters, esdf/mesh accumulators.
    esdf_lists[i] (optional): the block list update i walks, for vbx_esdf_update_blocks — the order of the reference's own
    container; without it vbx_esdf_update lists the blocks itself."""
    from voxblox_amd import capi  # noqa: F401
    acc = {"stage": {}, "counters": {}, "esdf_ms": 0.0, "esdf_cnt": {}, "mesh_s": 0.0, "mesh_blocks": 0, "mesh_vertices": 0,
           "esdf_each_ms": []}
    total = warmup + steps
    timed = [False]
    if esdf_cfg is not None:
        gm.enable_timing(True)      # per-update device times of the warm-up frames too (the first update of a map)
        # the integrator's workspace is reserved where the reference constructs its EsdfIntegrator (vbx_esdf_reserve), outside the
        # first update; what the reservation cost is reported next to it (esdf.workspace_reserve_ms)
        import torch
        torch.cuda.synchronize()
        _t0 = time.perf_counter()
        gm.esdf_reserve(esdf_cfg)
        acc["esdf_reserve_ms"] = (time.perf_counter() - _t0) * 1e3

    def step(i):
        pose, dp, dc, n = d_frames[i % len(d_frames)]
        gm.integrate_device(kind, cfg, pose[0], pose[1], dp.data_ptr(), dc.data_ptr(), n)
        if timed[0]:
            for k, v in gm.timing().items():
                acc["stage"][k] = acc["stage"].get(k, 0.0) + v
            for k, v in gm.counters().items():
                acc["counters"][k] = acc["counters"].get(k, 0) + v
        if esdf_cfg is not None:
            if esdf_lists is not None:
                gm.esdf_update_blocks(esdf_cfg, esdf_lists[i], incremental=True)
                acc["esdf_each_ms"].append(gm.timing()["total_ms"])
                gm.clear_updated(capi.UPDATE_ESDF, capi.LAYER_TSDF)
            else:
                gm.esdf_update(esdf_cfg, batch=False, clear_updated_flag=True)
                acc["esdf_each_ms"].append(gm.timing()["total_ms"])
            if timed[0]:
                acc["esdf_ms"] += acc["esdf_each_ms"][-1]
                for k, v in gm.counters().items():
                    if k.startswith("esdf"):
                        acc["esdf_cnt"][k] = acc["esdf_cnt"].get(k, 0) + v
        if mesh_cfg is not None:
            t0 = time.perf_counter()
            midx, moff = gm.mesh_generate(mesh_cfg, True, True, download=False)
            if timed[0]:
                acc["mesh_s"] += time.perf_counter() - t0
                acc["mesh_blocks"] += len(midx)
                acc["mesh_vertices"] += int(moff[-1])

    for i in range(warmup):
        step(i)
    gm.enable_timing(True)
    timed[0] = True
    barrier()
    t0 = time.perf_counter()
    for i in range(warmup, total):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    timed[0] = False
    pts = sum(d_frames[i % len(d_frames)][3] for i in range(warmup, total))
    return dt, pts, acc, step


SENSOR_BANDS = [1]   # ray bundles per sensor frame (--bands); whole sensors reproduce the reference's map


def sharded_stream_timed(sharded, d_frames, kind, cfg, warmup, total, barrier, dist, world, reduce_device):
    """The N > 1 timed region of the stream workload: W untimed frames per rank into the sharded map (delta map + pipelined
    owner exchange), a flush, then EXACTLY total - W timed frames bracketed by barrier + synchronize on both sides, and the
    MAX of the ranks' times (the contract's `value` = all ranks' points / that).  Every rank returns the same number.
    A function of its own so that tests/test_multi_gpu_gloo.py can run it at world size 8 over gloo with a CPU-backed map
    (deadlocks, ordering of the collectives, the MAX) where no 8-GPU box is at hand."""
    import torch
    for i in range(warmup):
        pose, dp, dc, n = d_frames[i % len(d_frames)]
        sharded.integrate_shard(kind, cfg, pose[0], pose[1], dp, dc, n)
    sharded.flush()
    for key in sharded.stats:
        sharded.stats[key] = 0
    barrier()
    t0 = time.perf_counter()
    for i in range(warmup, total):
        pose, dp, dc, n = d_frames[i % len(d_frames)]
        sharded.integrate_shard(kind, cfg, pose[0], pose[1], dp, dc, n)
    sharded.flush()
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], device=reduce_device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


def _bands(world):
    return max(SENSOR_BANDS[0], (max(world, 1) + 3) // 4)


def sensors4_shards(step, rank, world, dev, cache):
    """This rank's ray shards of one time step: 4 sensors x B bands, B = max(1, world / 4), dealt out in
    order (sensor-major), so world 1 holds everything, world 4 one sensor each, world 8 half a sensor each."""
    import torch
    from voxblox_amd import multi_gpu, scenes
    out = []
    for s, b, bands in multi_gpu.deal_sensor_units(world, bands=_bands(world))[rank]:
        key = (s, step % 25)
        if key not in cache:
            pose, pts, col = scenes.room_sensor_frame(s, step % 25)
            cache[key] = (pose, torch.from_numpy(pts).to(dev), torch.from_numpy(col).to(dev), pts.shape[0])
        pose, dp, dc, n = cache[key]
        lo, hi = multi_gpu.band_of(n, b, bands)
        out.append((pose[0], pose[1], dp[lo:hi], dc[lo:hi], hi - lo))
    return out


def run_sensors4(args, voxel, world, rank, local_rank, dist, dev, steps, warmup, barrier_fn, profile=True):
    import torch
    from voxblox_amd import capi, multi_gpu
    max_blocks = args.max_blocks or int(8192 * max(1.0, (VOXEL / voxel) ** 3) / 4)
    cfg = capi.tsdf_cfg(default_truncation_distance=4 * voxel, fast_observed_set=args.fast_set)
    kind = capi.TSDF_FAST
    pm = capi.Map(voxel, 16, max_blocks=max_blocks, device=local_rank)
    # one delta map per ray shard of this rank (x 2: the exchange of step k runs behind the integration of step k + 1):
    # the shards of a step are integrated concurrently, one host thread and one HIP stream each
    n_units = max(1, len(multi_gpu.deal_sensor_units(world, bands=_bands(world))[rank]))
    delta_sets = [[capi.Map(voxel, 16, max_blocks=max(2048, max_blocks // n_units), device=local_rank) for _ in range(n_units)]
                  for _ in range(2)]
    deltas = [delta_sets[0][0], delta_sets[1][0]]
    sharded = multi_gpu.PipelinedShardedTsdfMap(multi_gpu.GpuBackend(pm, dev),
                                                [[multi_gpu.GpuBackend(d, dev, keep_slots=True) for d in ds] for ds in delta_sets],
                                                rank, world, dist if world > 1 or os.environ.get("VBX_FORCE_COLLECTIVES") else None,
                                                device=dev)
    cache = {}
    for k in range(min(warmup + steps, 25)):
        sensors4_shards(k, rank, world, dev, cache)     # synthetic frames generated and uploaded before the clock
    torch.cuda.synchronize()
    use_native = args.exchange == "native" or (args.exchange == "auto" and world == 1)
    if use_native:
        # the C++ host path: libvbx_shard.so calls RCCL itself; torch.distributed only carries the communicator id
        from voxblox_amd import shard_native
        ids = [shard_native.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(ids, src=0)
        ns = shard_native.NativeShard(pm, delta_sets[0][0], rank, world, ids[0] if world > 1 else None, local_rank)
        for d in delta_sets[0][1:] + delta_sets[1]:
            ns.add_delta(d)
        ns.set_pipelined(True)

        def one(k):
            ns.begin_step()
            ns.integrate_shards(kind, cfg, [(pos, quat, dp.data_ptr(), dc.data_ptr(), n)
                                            for pos, quat, dp, dc, n in sensors4_shards(k, rank, world, dev, cache)])
            ns.end_step()

        for k in range(warmup):
            one(k)
        ns.wait()
        barrier_fn()
        t0 = time.perf_counter()
        for k in range(warmup, warmup + steps):
            one(k)
        ns.wait()
        barrier_fn()
        dt = time.perf_counter() - t0
        st = ns.stats()
        f = max(st["steps"], 1)
        exch = {"path": "libvbx_shard.so (C++, RCCL all-to-all-v called directly; shards integrated concurrently, one delta map "
                        "each; exchange on a worker thread behind the next step's integration)",
                "payload_bytes_per_step": int(st["payload_bytes"] / f), "sent_blocks_per_step": round(st["sent_blocks"] / f, 1)}
        ns.close()
    else:
        def barrier():
            sharded.flush()
            barrier_fn()

        for k in range(warmup):
            sharded.integrate_shards(kind, cfg, sensors4_shards(k, rank, world, dev, cache))
        sharded.flush()
        for key in sharded.stats:
            sharded.stats[key] = 0
        barrier()
        t0 = time.perf_counter()
        for k in range(warmup, warmup + steps):
            sharded.integrate_shards(kind, cfg, sensors4_shards(k, rank, world, dev, cache))
        barrier()
        dt = time.perf_counter() - t0
        f = max(sharded.stats["frames"], 1)
        exch = {"path": "voxblox_amd.multi_gpu (Python host path over torch.distributed)",
                "payload_bytes_per_step": int(sharded.stats["payload_bytes"] / f), "sent_blocks_per_step": round(sharded.stats["sent_blocks"] / f, 1),
                "integrate_ms_per_step": round(sharded.stats["integrate_s"] / f * 1e3, 3),
                "exchange_ms_per_step": round(sharded.stats["exchange_s"] / f * 1e3, 3),
                "wait_ms_per_step": round(sharded.stats["wait_s"] / f * 1e3, 3),
                "note": "this rank's figures; exchange = export of the touched blocks' sums + all-to-all to the owners + owner "
                        "merge, pipelined behind the next step's integration (its time is hidden unless wait_ms > 0)"}
    # per-kernel profile of the integration (delta map 0), outside the clock
    rows, calls = [], 0
    alg = {}
    if profile and args.profile_frames > 0 and rank == 0:
        d = deltas[0]
        d.profile(True, reset=True)
        n_prof = max(1, min(args.profile_frames // 4, 3))
        units = 0
        upd = 0
        for k in range(warmup + steps, warmup + steps + n_prof):
            d.clear()
            for pos, quat, dp, dc, n in sensors4_shards(k, rank, world, dev, cache):
                d.integrate_device(kind, cfg, pos, quat, dp.data_ptr(), dc.data_ptr(), n)
                units += n
                upd += d.counters()["voxels_touched"]
        d.profile(False)
        rows, calls = kernel_table(d, calls_per_step=calls_per_step_of(d, n_prof))
        alg = {"points": units / n_prof, "voxels_touched": upd / n_prof}
    sharded.close()
    return dt, exch, rows, alg, (pm, deltas)


def calls_per_step_of(gm, n_steps):
    _, calls = gm.profile_table()
    return max(calls / max(n_steps, 1), 1e-9)


# ------------------------------------------------------------------------------------------------
# the ONE line the driver parses: numbers only, < 4 KB; everything else goes to bench_detail.json
# ------------------------------------------------------------------------------------------------
LINE_LIMIT = 4096


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _short_roofline(r):
    if not isinstance(r, dict):
        return None
    o = _pick(r, ("bound", "achieved", "peak", "unit", "frac"))
    o["traffic"] = r.get("traffic")
    o.update(_pick(r, ("traffic_per_launch", "traffic_all_kernels_per_step", "traffic_over_algorithmic",
                       "traffic_all_kernels_over_algorithmic", "kernel", "launches_per_step", "avg_launch_us", "kernel_us_per_step",
                       "algorithmic_bytes_per_step", "step_frac", "device_ms_per_step")))
    td = r.get("traffic_detail")
    if isinstance(td, dict):
        o["traffic_source"] = td.get("source")
    return o


def _short_cpu(c):
    if not isinstance(c, dict):
        return None
    o = _pick(c, ("value", "unit", "cores", "kind", "host_hw_threads", "ms_per_update", "ms_per_step"))
    if isinstance(c.get("host"), dict):
        o.update(_pick(c["host"], ("affinity_threads", "cgroup_cpu_quota_cores")))
        o["cgroup_cpu_quota_cores"] = c["host"].get("cgroup_cpu_quota_cores")   # (null = unlimited: say so)
    if isinstance(c.get("by_threads"), dict):
        o["by_threads"] = {k: (v.get("value") if isinstance(v, dict) else v) for k, v in c["by_threads"].items()}
    if c.get("sample"):
        o["sample"] = str(c["sample"])[:64]
    return o


def _short_esdf(e):
    if not isinstance(e, dict):
        return None
    o = _pick(e, ("mode", "ms_per_update", "median_ms", "voxels_compared", "voxels_differing_from_reference",
                  "first_update_ms", "workspace_reserve_ms", "batch_update_ms"))
    if isinstance(e.get("cpu_baseline"), dict):
        o["cpu_1core"] = _pick(e["cpu_baseline"], ("ms_per_update", "first_update_ms", "batch_update_ms", "tsdf_ms_per_frame", "kind"))
    if isinstance(e.get("roofline"), dict):
        o["roofline"] = _pick(e["roofline"], ("achieved", "frac", "kernel", "kernel_us_per_step", "launches_per_step",
                                              "algorithmic_bytes_per_step", "traffic"))
    if isinstance(e.get("order_free"), dict):
        o["order_free_not_bit_exact"] = _pick(e["order_free"], ("ms_per_update", "value", "frac_gt_1e-4_m"))   # (rmse / max: bench_detail.json)
    return o


def _short_leg(j):
    if not isinstance(j, dict):
        return j
    if "error" in j:
        return {"error": str(j["error"])[:120]}
    o = _pick(j, ("value", "unit", "ms_per_step", "steps"))
    cfgd = j.get("config") or {}
    o.update(_pick(cfgd, ("ray_bundles_per_step", "semantics")))
    if "semantics" in o:
        o["semantics"] = str(o["semantics"])[:90]
    if isinstance(j.get("roofline"), dict):
        o["roofline"] = _pick(j["roofline"], ("achieved", "frac", "kernel", "kernel_us_per_step"))
    if isinstance(j.get("cpu_baseline"), dict):
        o["cpu_baseline"] = _pick(j["cpu_baseline"], ("value", "cores", "kind"))
    if isinstance(j.get("esdf"), dict):
        o["esdf"] = _short_esdf(j["esdf"])
    if isinstance(j.get("exchange"), dict):
        o["exchange"] = _pick(j["exchange"], ("payload_bytes_per_step", "sent_blocks_per_step"))
    for k in ("different_map_16_bundles", "whole_sensor_bundles"):
        if isinstance(j.get(k), dict):
            o[k] = _pick(j[k], ("value", "ms_per_step", "ray_bundles_per_step", "payload_bytes_per_step"))
    return o


def compact_line(out, detail_name):
    """The driver-facing line: BASELINE's metric with `roofline` and `cpu_baseline`, one `value` per secondary leg."""
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                       "vs_baseline", "dtype", "data"))
    line["vs_baseline"] = out.get("vs_baseline")
    cfgd = out.get("config") or {}
    line["config"] = _pick(cfgd, ("workload", "entry", "points_per_step", "points_per_step_per_gpu", "voxel_size", "voxels_per_side",
                                  "world_size_seen", "ray_bundles_per_step", "parallelism"))
    if "entry" in line["config"]:
        line["config"]["entry"] = str(line["config"]["entry"])[:64].rstrip(" :(") 
    for k in ("workload", "parallelism"):
        if k in line["config"]:
            line["config"][k] = str(line["config"][k])[:160]
    line["roofline"] = _short_roofline(out.get("roofline"))
    line["cpu_baseline"] = _short_cpu(out.get("cpu_baseline"))
    for k in ("host_pointer_path", "dropin_path"):
        if isinstance(out.get(k), dict):
            line[k] = _pick(out[k], ("value", "unit", "ms_per_step", "error", "entry"))
            if "entry" in line[k]:
                line[k]["entry"] = str(line[k]["entry"])[:110]
            if isinstance(out[k].get("ms_by_timer_tag"), dict):
                line[k]["ms_by_timer_tag"] = out[k]["ms_by_timer_tag"]
            if isinstance(out[k].get("by_host_blocks"), dict):   # host blocks -> [Mpoints/s, reconcile ms]
                line[k]["by_host_blocks"] = {hb: ([v.get("value"), (v.get("ms_by_timer_tag") or {}).get("hip/tsdf_reconcile_from_host")]
                                                  if "error" not in v else v) for hb, v in out[k]["by_host_blocks"].items()}
    if isinstance(out.get("esdf"), dict):
        line["esdf"] = _short_esdf(out["esdf"])
    if isinstance(out.get("exchange"), dict):
        line["exchange"] = _pick(out["exchange"], ("payload_bytes_per_step", "sent_blocks_per_step", "exchange_ms_per_step",
                                                   "wait_ms_per_step"))
    for k in ("different_map_16_bundles",):
        if isinstance(out.get(k), dict):
            line[k] = _pick(out[k], ("value", "ms_per_step", "ray_bundles_per_step", "payload_bytes_per_step"))
    if isinstance(out.get("n1_same_workload"), dict):
        line["n1_same_workload"] = out["n1_same_workload"].get("command")
    if isinstance(out.get("other_configs"), dict):
        line["legs"] = {str(k)[:40]: _short_leg(v) for k, v in out["other_configs"].items()}
    line["detail"] = detail_name
    text = json.dumps(line, separators=(",", ":"))
    # belt and braces: shed the optional parts, least important first, until the line fits
    def legs_drop(key):
        for leg in (line.get("legs") or {}).values():
            if isinstance(leg, dict):
                leg.pop(key, None)

    def legs_bare():
        line["legs"] = {k: _pick(v, ("value", "ms_per_step", "error")) for k, v in (line.get("legs") or {}).items()}

    shed = [lambda: legs_drop("roofline"), lambda: (line.get("dropin_path") or {}).pop("ms_by_timer_tag", None),
            lambda: [(line.get(k) or {}).pop("entry", None) for k in ("host_pointer_path", "dropin_path")],
            lambda: (line.get("dropin_path") or {}).pop("by_host_blocks", None),
            lambda: legs_drop("cpu_baseline"), lambda: line.pop("host_pointer_path", None),
            lambda: (line.get("cpu_baseline") or {}).pop("by_threads", None), lambda: legs_drop("esdf"), legs_bare,
            lambda: line.pop("esdf", None), lambda: line.pop("exchange", None), lambda: line.pop("legs", None)]
    for f in shed:
        if len(text) < LINE_LIMIT:
            break
        f()
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) < LINE_LIMIT, len(text)
    return text


def emit(out, args):
    """Full result -> bench_detail.json (and gpurun_out/ when that exists); the compact line -> stdout, LAST."""
    path = args.detail_out or os.path.join(ROOT, "bench_detail.json")
    try:
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
        scratch = os.path.join(ROOT, "gpurun_out")
        if not args.detail_out and os.path.isdir(scratch):
            with open(os.path.join(scratch, "bench_detail.json"), "w") as f:
                json.dump(out, f, indent=1)
    except OSError as e:
        sys.stderr.write(f"bench.py: could not write {path}: {e}\n")
    sys.stderr.flush()
    print(compact_line(out, os.path.basename(path)), flush=True)


# ------------------------------------------------------------------------------------------------
def main():
    args = parse()
    if args.cpu_threads:
        CPU_THREADS[:] = [int(x) for x in args.cpu_threads.split(",") if x.strip()]
    if args.gpus > 1 and "RANK" not in os.environ:
        spawn(args)
    import torch
    import torch.distributed as dist
    from voxblox_amd import capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force_sharded = bool(os.environ.get("VBX_FORCE_SHARDED")) and "RANK" in os.environ
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # test hook (tests/test_gpu_multi_merge.py): several ranks on ONE GPU over gloo, so that the N > 1 control flow of
    # this file — shard dealing, pipelined exchange, barriers, the MAX over ranks — runs on a one-GPU box (RCCL refuses
    # two ranks on one device).  Never set by the driver.
    one_gpu_gloo = os.environ.get("VBX_BENCH_ONE_GPU_GLOO") == "1"
    if one_gpu_gloo:
        local_rank = 0
    if world > 1 or force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu_gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if "RANK" in os.environ and world != args.gpus and rank == 0:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks; reporting n_gpus = {world}\n")

    workload = args.workload
    if workload == "auto":
        # every N runs the configuration the metric is quoted on (configs[1]): at N > 1 one sensor stream per GPU into ONE
        # map distributed by block owner (weak scaling: per-GPU work fixed), with configs[4] as a short secondary leg
        workload = "stream"
    voxel = float(args.voxel) or (0.02 if workload == "sensors4" else VOXEL)
    trunc = 4 * voxel
    steps = args.steps or (25 if workload == "sensors4" else 60)
    warmup = args.warmup if args.warmup >= 0 else (2 if workload == "sensors4" else 5)

    def barrier():
        if world > 1 or force_sharded:
            dist.barrier()
        torch.cuda.synchronize()

    def finish(out):
        if rank == 0:
            emit(out, args)
        if world > 1 or force_sharded:
            dist.barrier()
            dist.destroy_process_group()

    base = {"metric": "Mpoints/s integrated (640x480 frame, %g m voxels) + achieved HBM GB/s" % voxel,
            "unit": "Mpoints/s", "n_gpus": world, "steps": steps, "warmup": warmup, "higher_is_better": True,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic"}

    # ---------------------------------------------------------------------------- configs[4]
    if workload == "sensors4":
        SENSOR_BANDS[0] = max(1, args.bands or 1)
        nb = 4 * _bands(world)
        dt, exch, rows, alg, _maps = run_sensors4(args, voxel, world, rank, local_rank, dist, dev, steps, warmup,
                                                  lambda: barrier())
        if world > 1:
            tt = torch.tensor([dt], device=("cpu" if dist.get_backend() == "gloo" else dev), dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        pts_step = 4 * 307200
        out = dict(base)
        out.update({"value": round(pts_step * steps / dt / 1e6, 3), "ms_per_step": round(dt / steps * 1e3, 4), "scaling": "strong",
                    "config": {"workload": "BASELINE configs[4]: FastTsdfIntegrator, 4 concurrent 640x480 synthetic room sensors, "
                                           "%g m voxels / 16^3 blocks, trunc %g m; one step = all four frames (1,228,800 points)" % (voxel, trunc),
                               "points_per_step": pts_step, "voxel_size": voxel, "voxels_per_side": 16, "world_size_seen": world,
                               "semantics": ("whole-sensor ray bundles: the layout that reproduces the reference's map in the first step and stays "
                                             "within 2 % of the voxels afterwards (profiles/r04_shard_divergence_0p02.json)" if nb == 4 else
                                             "%d bundles per sensor: A DIFFERENT MAP than the reference integrates (every band restarts the "
                                             "Fast integrator's lossy sets); bit-exact only against an oracle doing the same shard + merge" % (nb // 4)),
                               "ray_bundles_per_step": nb,
                               "parallelism": ("%d ray bundle(s) per sensor = %d per step; " % (nb // 4, nb) +
                                               ("1 GPU integrates all of them concurrently (same shard + merge, no collective)" if world == 1 else
                                                f"{world} ranks, sparse RCCL all-to-all-v of touched blocks to their owners pipelined behind the next "
                                                "step's integration, map distributed by block owner"))},
                    "exchange": exch})
        if world > 1:
            # the driver's N = 1 run of `bench.py` is configs[1] (the metric's own configuration), a different workload:
            # the one-GPU point of THIS curve is the same shard + merge on one GPU
            out["n1_same_workload"] = {"command": "python bench.py --gpus 1 --workload sensors4" + (" --bands %d" % args.bands if args.bands else ""),
                                       "note": "strong-scaling efficiency at N = value / (N x that command's value); do not divide by the configs[1] line"}
        if rank == 0 and rows:
            alg_bytes = 16.0 * alg["points"] + 24.0 * alg["voxels_touched"]
            dev_ms = sum(r["us_per_step"] for r in rows) / 1e3
            out["roofline"] = roofline_from(rows, alg_bytes, dev_ms, "16 B x points + 24 B x distinct voxels updated, this rank's shards of one step")
            out["kernels"] = rows[:14]
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            from voxblox_amd import scenes
            fr = [scenes.room_sensor_frame(0, k) for k in range(8)]
            out["cpu_baseline"] = cpu_baseline(fr, "fast", voxel, args.cpu_seconds)
        if world == 1 and nb == 4 and not args.bands and rank == 0:
            # rounds 3-4 quoted this layout; it integrates ANOTHER map than the reference, so it is a secondary figure
            try:
                del _maps
                torch.cuda.empty_cache()
                SENSOR_BANDS[0] = 4
                dt16, exch16, _r, _a, _m = run_sensors4(args, voxel, world, rank, local_rank, dist, dev, steps, warmup, lambda: barrier(), profile=False)
                out["different_map_16_bundles"] = {"value": round(pts_step * steps / dt16 / 1e6, 3), "ms_per_step": round(dt16 / steps * 1e3, 4),
                                                   "ray_bundles_per_step": 16, "payload_bytes_per_step": exch16.get("payload_bytes_per_step"),
                                                   "note": "four bands per sensor: not the reference's map (12 % of voxels > 1e-4 m, half the weight missing)"}
            except Exception as e:
                out["different_map_16_bundles"] = {"error": repr(e)[:200]}
        finish(out)
        return

    # ---------------------------------------------------------------------------- stream (configs[1..3])
    kind = {"fast": capi.TSDF_FAST, "merged": capi.TSDF_MERGED, "simple": capi.TSDF_SIMPLE}[args.integrator]
    total = warmup + steps
    frames = stream_frames(args.scene, rank, total + args.profile_frames + args.mirror_frames + 8, world)
    d_frames = to_device(frames, dev)
    n_pts = frames[0][1].shape[0]
    max_blocks = args.max_blocks or int(8192 * max(1.0, (VOXEL / voxel) ** 3))
    cfg = capi.tsdf_cfg(default_truncation_distance=trunc, merged_bundle_order=args.merged_order, fast_observed_set=args.fast_set)
    ecfg = capi.esdf_cfg(min_distance_m=trunc / 2, reference_order=0) if args.esdf else None  # ros_params.h:136-137 (order_free mode)
    mcfg = capi.mesh_cfg() if args.mesh else None

    sharded = None
    if world > 1 or force_sharded:
        # weak scaling variant: one sensor's stream per rank, delta map + sparse exchange per frame
        from voxblox_amd import multi_gpu
        pm = capi.Map(voxel, 16, max_blocks=max_blocks, device=local_rank)
        dl = [capi.Map(voxel, 16, max_blocks=max_blocks, device=local_rank) for _ in range(2)]
        sharded = multi_gpu.PipelinedShardedTsdfMap(multi_gpu.GpuBackend(pm, dev), [multi_gpu.GpuBackend(d, dev, keep_slots=True) for d in dl],
                                                    rank, world, dist, device=dev)
        dt = sharded_stream_timed(sharded, d_frames, kind, cfg, warmup, total, barrier, dist, world,
                                  "cpu" if dist.get_backend() == "gloo" else dev)
        pts_timed = sum(d_frames[i % len(d_frames)][3] for i in range(warmup, total))
        f = max(sharded.stats["frames"], 1)
        out = dict(base)
        out.update({"value": round(world * pts_timed / dt / 1e6, 3), "ms_per_step": round(dt / steps * 1e3, 4), "scaling": "weak",
                    "config": {"workload": f"BASELINE configs[1] on every GPU: {args.integrator.capitalize()}TsdfIntegrator, one 640x480 synthetic "
                                           f"{args.scene} scan stream per rank (the ranks' sensors move through the same room, a quarter "
                                           "turn apart), %g m voxels / 16^3 blocks, trunc %g m; one step = one frame per rank" % (voxel, trunc),
                               "points_per_step": n_pts * world, "points_per_step_per_gpu": n_pts, "voxel_size": voxel, "voxels_per_side": 16,
                               "world_size_seen": world,
                               "semantics": "ray-bundle sharding by sensor: every rank integrates its cloud into a zeroed delta map (bit-exact "
                                            "integrator per shard); touched blocks go to their owner ranks, merged there with "
                                            "mergeVoxelAIntoVoxelB semantics into ONE map distributed by block owner",
                               "parallelism": f"{world} sensors, one per GPU; sparse RCCL all-to-all-v of the touched blocks' weighted sums, "
                                              "pipelined behind the next frame's integration"},
                    "exchange": {"payload_bytes_per_step": int(sharded.stats["payload_bytes"] / f),
                                 "exchange_ms_per_step": round(sharded.stats["exchange_s"] / f * 1e3, 3),
                                 "integrate_ms_per_step": round(sharded.stats["integrate_s"] / f * 1e3, 3),
                                 "wait_ms_per_step": round(sharded.stats["wait_s"] / f * 1e3, 3),
                                 "note": "this rank's figures; the exchange is hidden behind the next frame unless wait_ms > 0"}})
        # per-kernel profile of THIS rank's integration (rank 0 only, outside the clock): its delta map 0 over the timed frame
        # indices, an event pair around every launch -> the dominant kernel and its share of the HBM roof, like at N = 1
        if rank == 0 and args.profile_frames > 0:
            try:
                sharded.flush()
                gp = dl[0]
                gp.enable_timing(False)
                gp.profile(True, reset=True)
                upd = pts_prof = 0
                nprof = min(steps, max(4, args.profile_frames))
                for i in range(warmup, warmup + nprof):
                    pose, dp, dc, n = d_frames[i % len(d_frames)]
                    gp.clear_keep_slots()
                    gp.integrate_device(kind, cfg, pose[0], pose[1], dp.data_ptr(), dc.data_ptr(), n)
                    upd += gp.counters()["voxels_touched"]
                    pts_prof += n
                gp.profile(False)
                rows, _ = kernel_table(gp, 1.0)
                alg_bytes = 16.0 * pts_prof / nprof + 24.0 * upd / nprof
                dev_ms = sum(r["us_per_step"] for r in rows) / 1e3
                out["roofline"] = roofline_from(rows, alg_bytes, dev_ms, "16 B x points + 24 B x distinct voxels updated per frame (SURVEY 8(d)), this rank's frames into its per-frame delta map")
                out["kernels"] = rows[:16]
            except Exception as e:  # a secondary measurement must never take the line down
                out["roofline_error"] = repr(e)[:200]
        sharded.close()
        del sharded, pm, dl
        torch.cuda.empty_cache()
        if world > 1 and not args.no_extras:
            # BASELINE configs[4] as a secondary leg of the same launch: the four 0.02 m sensors' frames of a step dealt
            # out over the ranks (strong scaling: the work per step is fixed)
            try:
                v4 = float(args.voxel) or 0.02
                s4, w4 = 4, 2  # two warm-up steps: each of the two alternating delta sets has been used once (its buffers exist)
                dt4, exch4, _rows, _alg, _maps = run_sensors4(args, v4, world, rank, local_rank, dist, dev, s4, w4, lambda: barrier(), profile=False)
                t4 = torch.tensor([dt4], device=("cpu" if dist.get_backend() == "gloo" else dev), dtype=torch.float64)
                dist.all_reduce(t4, op=dist.ReduceOp.MAX)
                dt4 = float(t4.item())
                out["other_configs"] = {"configs[4]: 4 sensors, %g m, ray shards dealt over the ranks (strong scaling)" % v4: {
                    "value": round(4 * 307200 * s4 / dt4 / 1e6, 3), "unit": "Mpoints/s", "ms_per_step": round(dt4 / s4 * 1e3, 4), "steps": s4,
                    "points_per_step": 4 * 307200, "exchange": exch4,
                    "one_gpu_same_workload": "profiles/r03_bench_sensors4_1gpu.json: 32.5 Mpoints/s, 37.8 ms per step (sixteen bundles on one GPU through libvbx_shard.so; 43-55 ms through the Python host path this leg uses)"}}
            except Exception as e:  # a secondary leg must never take the headline line down
                out["other_configs"] = {"configs[4]": {"error": repr(e)}}
        finish(out)
        return

    gm = capi.Map(voxel, 16, max_blocks=max_blocks, device=local_rank)
    gm.set_stream(torch.cuda.current_stream().cuda_stream)
    esdf_ref = None
    esdf_parity = args.esdf and args.esdf_mode == "reference"
    if esdf_parity:
        # the timed region runs the reference's OWN result (vbx_esdf_cfg.reference_order = 1).  The reference build walks the
        # same frames first (outside the clock): its times are the CPU figures of this leg, its final layer is what the device's
        # layer is compared with afterwards.
        ecfg = capi.esdf_cfg(min_distance_m=trunc / 2, reference_order=1)
        if not args.no_cpu_baseline:
            esdf_ref = esdf_reference_run(frames, voxel, total)
    use_lists = esdf_ref is not None and os.environ.get("VBX_BENCH_ESDF_LISTS", "0") == "1"
    dt, pts_timed, acc, step = run_stream(args, gm, kind, cfg, d_frames, steps, warmup, barrier, ecfg, mcfg,
                                          esdf_lists=esdf_ref["lists"] if use_lists else None)
    K = steps
    workload_name = {("fast", "room"): "BASELINE configs[1]", ("merged", "cow"): "BASELINE configs[2]"}.get((args.integrator, args.scene), "variant")
    if args.esdf and args.integrator == "fast" and args.scene == "room":
        workload_name = "BASELINE configs[3]"
    out = dict(base)
    out.update({"value": round(pts_timed / dt / 1e6, 3), "ms_per_step": round(dt / K * 1e3, 4), "scaling": "weak",
                "config": {"workload": f"{workload_name}: {args.integrator.capitalize()}TsdfIntegrator, 640x480 synthetic "
                                       + ("room scan stream" if args.scene == "room" else "Cow-and-Lady-style orbit (room + sphere + cylinder, 10 % pixels dropped)")
                                       + (", EsdfIntegrator::updateFromTsdfLayer(true) after every frame" if args.esdf else "")
                                       + (", MeshIntegrator::generateMesh(true, true) after every frame" if args.mesh else "")
                                       + ", %g m voxels / 16^3 blocks, trunc %g m" % (voxel, trunc),
                           "points_per_step": n_pts, "voxel_size": voxel, "voxels_per_side": 16, "world_size_seen": world,
                           "scene": args.scene, "esdf_after_each_frame": bool(args.esdf), "mesh_after_each_frame": bool(args.mesh),
                           "semantics": ("fast mode (merged_bundle_order=%d, fast_observed_set=%d)" % (args.merged_order, args.fast_set)
                                         if (args.merged_order or args.fast_set) else
                                         "TSDF bit-exact vs the 1-thread reference; ESDF order-free fixed point, NOT bit-exact" if (args.esdf and not esdf_parity)
                                         else "bit-exact vs the 1-thread reference"),
                           "entry": "vbx_tsdf_integrate_device: points and colours resident in HBM when the clock starts (the bench "
                                    "contract); the same frames through voxblox's own FastTsdfIntegrator::integratePointCloud with the "
                                    "drop-in linked in (host Layer, host pointers — SURVEY 8(d)'s definition of the metric) = dropin_path",
                           "parallelism": "1 GPU, whole cloud"}})
    stage = {k: round(v / K, 4) for k, v in acc["stage"].items()}
    out["stage_ms"] = stage
    out["counters_per_step"] = {k: round(v / K, 1) for k, v in acc["counters"].items() if not k.startswith("esdf")}

    # ---- per-kernel profile pass (outside the clock) over the SAME frames the clock ran over: a second map, the
    # warm-up frames unprofiled, then the timed frame indices with an event pair around every launch
    if args.profile_frames > 0:
        P = steps
        gp = capi.Map(voxel, 16, max_blocks=max_blocks, device=local_rank)
        gp.set_stream(torch.cuda.current_stream().cuda_stream)
        gp.enable_timing(False)
        for i in range(warmup):
            pose, dp, dc, n = d_frames[i % len(d_frames)]
            gp.integrate_device(kind, cfg, pose[0], pose[1], dp.data_ptr(), dc.data_ptr(), n)
        gp.profile(True, reset=True)
        upd = 0
        pts_prof = 0
        for i in range(warmup, total):
            pose, dp, dc, n = d_frames[i % len(d_frames)]
            gp.integrate_device(kind, cfg, pose[0], pose[1], dp.data_ptr(), dc.data_ptr(), n)
            upd += gp.counters()["voxels_touched"]
            pts_prof += n
        gp.profile(False)
        rows, _ = kernel_table(gp, 1.0)
        gp.close()
        del gp
        alg_bytes = 16.0 * pts_prof / P + 24.0 * upd / P
        out["roofline"] = roofline_from(rows, alg_bytes, stage.get("total_ms", 0.0),
                                        "16 B x points + 24 B x distinct voxels updated per frame (SURVEY 8(d))")
        out["roofline"]["frames_profiled"] = [warmup, total - 1]
        out["kernels"] = rows[:16]
        if args.esdf:
            ge = capi.Map(voxel, 16, max_blocks=max_blocks, device=local_rank)
            ge.set_stream(torch.cuda.current_stream().cuda_stream)
            ge.enable_timing(False)
            blocks = relax = 0
            for i in range(total):
                pose, dp, dc, n = d_frames[i % len(d_frames)]
                ge.profile(False)
                ge.integrate_device(kind, cfg, pose[0], pose[1], dp.data_ptr(), dc.data_ptr(), n)
                if i == warmup:
                    ge.profile(True, reset=True)
                elif i > warmup:
                    ge.profile(True)
                if use_lists:
                    ge.esdf_update_blocks(ecfg, esdf_ref["lists"][i], incremental=True)
                    ge.clear_updated(capi.UPDATE_ESDF, capi.LAYER_TSDF)
                else:
                    ge.esdf_update(ecfg, batch=False, clear_updated_flag=True)
                if i >= warmup:
                    c = ge.counters()
                    blocks += c["esdf_blocks"]
                    relax += c["esdf_relaxations"]
            ge.profile(False)
            erows, _ = kernel_table(ge, 1.0)
            ge.close()
            del ge
            ealg = (52.0 * 4096 * blocks + 40.0 * relax) / P
            esdf_ms = acc["esdf_ms"] / K
            each = acc["esdf_each_ms"]
            out["esdf"] = {"mode": "reference_order=1 (the reference's own result)" if esdf_parity else "order-free fixed point, NOT bit-exact",
                           "ms_per_update": round(esdf_ms, 4), "median_ms": round(float(np.median(each[warmup:])), 4),
                           "first_update_ms": round(each[0], 3) if each else None,
                           "workspace_reserve_ms": round(acc.get("esdf_reserve_ms", 0.0), 3),
                           "counters_per_update": {k: round(v / K, 1) for k, v in acc["esdf_cnt"].items()},
                           "roofline": roofline_from(erows, ealg, esdf_ms, "52 B x 4096 x updated blocks + 40 B x successful relaxations (SURVEY 8(d))"),
                           "kernels": erows[:8],
                           "block_list": ("the reference container's own order, handed to vbx_esdf_update_blocks" if use_lists else
                                          "vbx_esdf_update lists the blocks itself" + (" in the order the reference's Layer would hold them" if esdf_parity else ""))}
    elif args.esdf:
        out["esdf"] = {"ms_per_update": round(acc["esdf_ms"] / K, 4)}
    gm.enable_timing(True)

    if args.mesh and K:
        out["mesh"] = {"ms_per_update": round(acc["mesh_s"] / K * 1e3, 4), "blocks_per_update": round(acc["mesh_blocks"] / K, 1),
                       "vertices_per_update": round(acc["mesh_vertices"] / K, 1),
                       "note": "host wall clock of vbx_mesh_generate, vertices left device-resident"}
        if not args.no_cpu_baseline:
            out["mesh"]["cpu_reference"] = cpu_mesh_baseline(frames[:12], args.integrator, voxel)

    # ---- host-mirror cost, outside the timed region
    if args.mirror_frames > 0 and not args.esdf and not args.mesh:
        staging = gm.pinned_voxels(2048)
        gm.clear_updated(capi.UPDATE_MAP)
        torch.cuda.synchronize()
        nb = nbytes = 0
        t_int = t_mir = 0.0
        base_i = total + 2 * args.profile_frames
        for j in range(args.mirror_frames):
            ta = time.perf_counter()
            pose, dp, dc, n = d_frames[(base_i + j) % len(d_frames)]
            gm.integrate_device(kind, cfg, pose[0], pose[1], dp.data_ptr(), dc.data_ptr(), n)
            torch.cuda.synchronize()
            tb = time.perf_counter()
            upd = gm.blocks_updated(capi.UPDATE_MAP)
            if len(upd) > staging.shape[0]:
                staging = gm.pinned_voxels(len(upd) * 5 // 4)
                tb = time.perf_counter()
            vox, bits, hd = gm.blocks_download(upd, out=staging)
            gm.clear_updated(capi.UPDATE_MAP)
            tc = time.perf_counter()
            t_int += tb - ta
            t_mir += tc - tb
            nb += len(upd)
            nbytes += vox.nbytes
        M = args.mirror_frames
        out["host_mirror"] = {"frames": M, "blocks_per_frame": round(nb / M, 1), "MB_per_frame": round(nbytes / M / 1e6, 3),
                              "integrate_ms": round(t_int / M * 1e3, 4), "mirror_ms": round(t_mir / M * 1e3, 4),
                              "note": "mirror = list updated blocks + AoS pack kernel + one D2H copy into page-locked staging + clear kMap bits"}

    # ---- the entry point voxblox callers use: vbx_tsdf_integrate with HOST pointers (Pointcloud::data(), pageable
    # memory), i.e. the same step plus the 16 B/point staging copy.  Same frames, same warm-up, a map of its own;
    # reported beside `value`, which by the bench contract times resident inputs.
    if not args.esdf and not args.mesh and not args.no_host_path:
        gh = capi.Map(voxel, 16, max_blocks=max_blocks, device=local_rank)
        for i in range(warmup):
            pose, pts, col = frames[i % len(frames)]
            gh.integrate(kind, cfg, pose[0], pose[1], pts, col)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(warmup, total):
            pose, pts, col = frames[i % len(frames)]
            gh.integrate(kind, cfg, pose[0], pose[1], pts, col)
        torch.cuda.synchronize()
        dth = time.perf_counter() - t0
        out["host_pointer_path"] = {"value": round(pts_timed / dth / 1e6, 3), "unit": "Mpoints/s", "ms_per_step": round(dth / K * 1e3, 4),
                                    "entry": "vbx_tsdf_integrate (host pointers)",
                                    "note": "vbx_tsdf_integrate (the drop-in's entry: pageable host points + colours copied to HBM "
                                            "inside the call), same frames and warm-up as `value`"}
        gh.close()
        del gh

    # ---- the drop-in itself: FastTsdfIntegrator::integratePointCloud of voxblox's own class with the HIP translation
    # units linked in (oracle/_ref/libvbxref_hip.so = the reference's headers + voxblox_amd/host/dropin/*.cc): host
    # Layer reconciled, points copied in, integrated on the device, touched blocks mirrored back into the host Layer
    if not args.esdf and not args.mesh and not args.no_host_path and not args.no_cpu_baseline:
        try:
            out["dropin_path"] = dropin_leg(frames, args.integrator, voxel, warmup, steps)
            # the same call with a host Layer that also holds 5 k / 20 k blocks the frames never touch (a map that has been
            # built for a while): what the call costs must follow the blocks a cloud touches, like the reference's, not the map
            by = {str(out["dropin_path"].get("host_blocks_at_end")): _pick(out["dropin_path"], ("value", "ms_per_step", "ms_by_timer_tag"))}
            for extra in DROPIN_EXTRA_BLOCKS:
                try:
                    j = dropin_leg(frames, args.integrator, voxel, warmup, steps, extra_blocks=extra)
                    by[str(j.get("host_blocks_at_end"))] = _pick(j, ("value", "ms_per_step", "ms_by_timer_tag"))
                except Exception as e:
                    by["+%d" % extra] = {"error": repr(e)[:200]}
            out["dropin_path"]["by_host_blocks"] = by
        except Exception as e:  # a secondary leg must never take the headline line down
            out["dropin_path"] = {"error": repr(e)[:300]}

    # ---- CPU reference on the same box
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(frames[:total], args.integrator, voxel, args.cpu_seconds,
                                           window=(warmup, steps) if voxel >= 0.049 else None)
        if args.esdf and esdf_ref is not None:
            # parity of the timed region itself: the device layer after the last timed frame against the reference's
            out["esdf"].update(esdf_layer_diff(gm, esdf_ref["map"].esdf_dict()))
            te, tt = esdf_ref["esdf_ms"], esdf_ref["tsdf_ms"]
            out["esdf"]["cpu_baseline"] = {"ms_per_update": round(float(np.median(te[warmup:])), 3), "first_update_ms": round(te[0], 3),
                                           "tsdf_ms_per_frame": round(float(np.median(tt[warmup:])), 3), "cores": 1, "kind": esdf_ref["kind"],
                                           "sample": "the frames of the timed region, reference build, 1 integrator thread (the deterministic reference), "
                                                     "updateFromTsdfLayer(true) after every frame"}
            # updateFromTsdfLayerBatch (esdf_integrator.cc:94-102) of the final TSDF map, both sides
            try:
                t0 = time.perf_counter()
                esdf_ref["esdf"].update_from_tsdf_layer_batch()
                out["esdf"]["cpu_baseline"]["batch_update_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
                gm.enable_timing(True)
                if use_lists:
                    gm.clear(capi.LAYER_ESDF)
                    gm.esdf_update_blocks(ecfg, esdf_ref["all_blocks"], incremental=False)
                else:
                    gm.esdf_update(ecfg, batch=True, clear_updated_flag=True)
                out["esdf"]["batch_update_ms"] = round(gm.timing()["total_ms"], 2)
                out["esdf"]["batch"] = esdf_layer_diff(gm, esdf_ref["map"].esdf_dict())
            except Exception as e:  # a secondary measurement must never take the line down
                out["esdf"]["batch"] = {"error": repr(e)[:200]}
            # the order-free mode on the same frames: fast, another answer
            try:
                go = capi.Map(voxel, 16, max_blocks=max_blocks, device=local_rank)
                go.set_stream(torch.cuda.current_stream().cuda_stream)
                ocfg = capi.esdf_cfg(min_distance_m=trunc / 2, reference_order=0)
                dto, ptso, acco, _ = run_stream(args, go, kind, cfg, d_frames, steps, warmup, barrier, ocfg, None)
                # (the reference map went through the batch update above: compare with a fresh incremental run)
                ref2 = esdf_reference_run(frames, voxel, total)
                dd = esdf_layer_diff(go, ref2["map"].esdf_dict())
                out["esdf"]["order_free"] = {"value": round(ptso / dto / 1e6, 3), "ms_per_update": round(acco["esdf_ms"] / K, 4),
                                             "rmse_m_vs_reference": dd["rmse_m"], "max_m_vs_reference": dd["max_abs_m"],
                                             "frac_gt_1e-4_m": dd["frac_gt_1e-4_m"], "voxels_differing_from_reference": dd["voxels_differing_from_reference"],
                                             "note": "vbx_esdf_cfg.reference_order = 0; NOT the reference's result"}
                go.close()
                del go, ref2
            except Exception as e:
                out["esdf"]["order_free"] = {"error": repr(e)[:200]}
        elif args.esdf:
            out["esdf"]["cpu_baseline"] = cpu_esdf_baseline(frames[:40], voxel, min(args.cpu_seconds, 12.0))
            if args.esdf_fidelity_frames > 0:
                nf = min(args.esdf_fidelity_frames, len(frames))
                out["esdf"]["fidelity"] = esdf_fidelity(frames, voxel, nf, {max(1, nf // 4), max(1, nf // 2), max(1, 3 * nf // 4)})

    # ---- default run: short secondary legs for the other single-GPU configs
    default_run = (args.integrator == "fast" and args.scene == "room" and not args.esdf and not args.mesh
                   and abs(voxel - VOXEL) < 1e-9 and args.fast_set == 0)
    if default_run and not args.no_extras:
        extras = {}
        py = [sys.executable, os.path.abspath(__file__), "--no-extras", "--no-host-path", "--mirror-frames", "0"]
        legs = {"configs[2] merged cow orbit": ["--integrator", "merged", "--scene", "cow", "--steps", "20", "--warmup", "3", "--cpu-seconds", "6"],
                "configs[3] fast + esdf (reference order)": ["--esdf", "--steps", "20", "--warmup", "3", "--cpu-seconds", "6"],
                "configs[4] 4 sensors 0.02 m, 1 GPU": ["--workload", "sensors4", "--steps", "4", "--warmup", "2"]}
        del gm
        torch.cuda.empty_cache()
        if args.cpu_threads:
            py += ["--cpu-threads", args.cpu_threads]
        if steps < 10:   # a short run (tests): the legs shorten themselves with it
            for extra in legs.values():
                if "--steps" in extra:
                    extra[extra.index("--steps") + 1] = str(steps)
                    extra[extra.index("--warmup") + 1] = str(min(warmup, 2))
        import tempfile
        for name, extra in legs.items():
            try:
                with tempfile.NamedTemporaryFile(suffix=".json", delete=False) as tf:
                    leg_path = tf.name
                r = subprocess.run(py + extra + ["--detail-out", leg_path], capture_output=True, text=True, timeout=900)
                if r.returncode == 0 and os.path.getsize(leg_path) > 2:
                    j = json.load(open(leg_path))
                    keep = {k: j[k] for k in ("value", "unit", "ms_per_step", "steps", "config", "cpu_baseline", "esdf", "exchange", "roofline",
                                              "different_map_16_bundles") if k in j}
                    extras[name] = keep
                else:
                    extras[name] = {"error": (r.stderr or r.stdout)[-300:]}
                os.unlink(leg_path)
            except Exception as e:  # a secondary leg must never take the headline line down
                extras[name] = {"error": repr(e)}
        out["other_configs"] = extras
    finish(out)


if __name__ == "__main__":
    main()
