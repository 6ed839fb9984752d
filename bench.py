#!/usr/bin/env python
"""bench.py — Mpoints/s integrated by the HIP TSDF hot path on MI355X.

Workload (BASELINE.json configs[1]): FastTsdfIntegrator, 640x480 synthetic room-scan stream
(voxblox_amd.scenes.room_frame), 0.05 m voxels / 16^3 blocks, truncation 4 voxels, all other
Config defaults.  One "step" = one integratePointCloud() call on one 307,200-point frame
whose points/colours are already resident in HBM (vbx_tsdf_integrate_device).

Contract (see the task statement): W untimed warm-up steps, then exactly K timed steps
bracketed by barrier + torch.cuda.synchronize() on both sides, MAX over ranks, rank 0 prints
ONE JSON line.  For --gpus N every rank integrates its own sensor's frame (weak scaling) into a per-frame
delta map; overlapping block updates are merged with an RCCL reduce-scatter into a persistent
map distributed by block ownership (voxblox_amd/multi_gpu.py).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

VOXEL = 0.05
TRUNC = 4 * VOXEL
N_STREAM = 100
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--integrator", default="fast", choices=["fast", "merged", "simple"])
    ap.add_argument("--esdf", action="store_true",
                    help="BASELINE configs[3]: EsdfIntegrator::updateFromTsdfLayer(true) after every frame")
    ap.add_argument("--mesh", action="store_true",
                    help="MeshIntegrator::generateMesh(only_mesh_updated_blocks=true, clear_updated_flag=true) "
                         "after every frame (SURVEY 8(f) #4), reported beside the integration")
    ap.add_argument("--scene", default="room", choices=["room", "cow"],
                    help="room = configs[1]/[3] stream; cow = configs[2] Cow-and-Lady-style orbit")
    ap.add_argument("--voxel", type=float, default=VOXEL,
                    help="voxel size (default = configs[1]'s 0.05 m; 0.02 = configs[4]'s resolution); "
                         "truncation stays 4 voxels")
    ap.add_argument("--max-blocks", type=int, default=0, help="block pool capacity (0 = sized from --voxel)")
    ap.add_argument("--merged-order", type=int, default=0, choices=[0, 1],
                    help="Merged only: 0 = the reference's unordered_map bundle order (bit-exact, host replay), "
                         "1 = ascending voxel key (no host step)")
    ap.add_argument("--fast-set", type=int, default=0, choices=[0, 1],
                    help="Fast only: 0 = the reference's approximate observed-voxel set (bit-exact, iterative replay), "
                         "1 = exact voxel set (one solve)")
    ap.add_argument("--no-variants", action="store_true", help="skip the fast-mode variant measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mirror-frames", type=int, default=8,
                    help="extra untimed-for-`value` frames that also mirror the touched blocks to the host "
                         "(vbx_blocks_updated + vbx_blocks_download + vbx_clear_updated); 0 = skip")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of the CPU sample (0 = auto)")
    return ap.parse_args()


def cpu_baseline(frames, kind, voxel=VOXEL):
    """Times the oracle (CPU restatement of the reference, reference threading scheme) on a
    bounded sample of the same stream: threads = host cores, median frame after 3 warm-ups."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ctypes
    import oracle_py as O
    cores = os.cpu_count() or 1
    # oracle/_ref = the reference's own sources (over dependency shims) when it was built;
    # otherwise the restatement.
    use_ref = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libvbxref.so"))
    L = O.ref_lib() if use_ref else O.lib()
    best = None
    for threads in sorted({1, min(cores, 8), cores}):
        L.orc_fast_reset_counter_set(0)
        m = O.OracleMap(voxel, 16, L=L)
        c = O.TsdfCfg()
        L.orc_tsdf_cfg_default(ctypes.byref(c))
        c.default_truncation_distance = 4 * voxel
        c.integrator_threads = threads
        it = m.tsdf_integrator(kind, c)
        ts = []
        t_begin = time.time()
        for i, (pose, pts, col) in enumerate(frames):
            t0 = time.perf_counter()
            it.integrate(pose[0], pose[1], pts, col)
            ts.append(time.perf_counter() - t0)
            if time.time() - t_begin > 8.0 and i >= 5:
                break
        used = ts[3:] if len(ts) > 4 else ts
        med = float(np.median(used))
        rec = dict(value=round(frames[0][1].shape[0] / med / 1e6, 3), threads=threads,
                   frames=len(ts), median_ms=round(med * 1e3, 2))
        if best is None or rec["value"] > best["value"]:
            best = rec
        del it, m
    return {"value": best["value"], "unit": "Mpoints/s", "cores": best["threads"],
            "kind": "reference" if use_ref else "port",
            "sample": f"{best['frames']} frames of the same 640x480 room stream, {kind} integrator "
                      + ("(reference sources compiled over dependency shims, oracle/_ref), "
                         if use_ref else "(oracle restatement), ")
                      + f"median frame {best['median_ms']} ms after 3 warm-up frames, "
                      f"best of integrator_threads in {{1,{min(cores, 8)},{cores}}} (host has {cores} hw threads)"}


def cpu_mesh_baseline(frames, kind, voxel):
    """The reference's MeshIntegrator (oracle/_ref when built, else the restatement) after every
    frame of a short sample of the same stream: generateMesh(true, true), 1 thread and all cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ctypes
    import oracle_py as O
    cores = os.cpu_count() or 1
    use_ref = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libvbxref.so"))
    L = O.ref_lib() if use_ref else O.lib()
    res = {}
    for threads in (sorted({1, cores}) if use_ref else [1]):   # the restatement meshes on one thread
        L.orc_fast_reset_counter_set(0)
        m = O.OracleMap(voxel, 16, L=L)
        c = O.TsdfCfg()
        L.orc_tsdf_cfg_default(ctypes.byref(c))
        c.default_truncation_distance = 4 * voxel
        c.integrator_threads = cores
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


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from voxblox_amd import capi, scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force_sharded = bool(os.environ.get("VBX_FORCE_SHARDED")) and "RANK" in os.environ
    if world > 1 or force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    kind = {"fast": capi.TSDF_FAST, "merged": capi.TSDF_MERGED, "simple": capi.TSDF_SIMPLE}[args.integrator]

    total = args.warmup + args.steps
    # Synthetic stream: rank r starts its sweep a quarter turn further (its own sensor).
    if args.scene == "cow":
        frames = [scenes.cow_and_lady_like_frame((k + 50 * rank) % 200) for k in range(min(total, 200))]
    else:
        frames = [scenes.room_frame((k + 25 * rank) % N_STREAM, N_STREAM) for k in range(min(total, N_STREAM))]
    d_frames = []
    for pose, pts, col in frames:
        d_frames.append((pose, torch.from_numpy(pts).to(dev), torch.from_numpy(col).to(dev)))
    n_pts = frames[0][1].shape[0]
    n_pts_all = [f[1].shape[0] for f in frames]

    voxel = float(args.voxel)
    trunc = 4 * voxel
    max_blocks = args.max_blocks or int(8192 * max(1.0, (VOXEL / voxel) ** 3))
    gm = capi.Map(voxel, 16, max_blocks=max_blocks, device=local_rank)
    cfg = capi.tsdf_cfg(default_truncation_distance=trunc, merged_bundle_order=args.merged_order,
                        fast_observed_set=args.fast_set)
    sharded = None
    if not (world > 1 or force_sharded):
        gm.set_stream(torch.cuda.current_stream().cuda_stream)
    else:  # every map keeps its own non-blocking stream so that integration and exchange overlap
        # Ray-bundle sharding: gm is this rank's per-frame delta map; the
        # persistent map is distributed by block ownership and fed by an RCCL reduce-scatter.
        from voxblox_amd import multi_gpu
        pm = capi.Map(voxel, 16, max_blocks=max_blocks, device=local_rank)
        gm2 = capi.Map(voxel, 16, max_blocks=max_blocks, device=local_rank)   # second delta map (double buffer)
        sharded = multi_gpu.PipelinedShardedTsdfMap(multi_gpu.GpuBackend(pm, dev),
                                                    [multi_gpu.GpuBackend(gm, dev), multi_gpu.GpuBackend(gm2, dev)],
                                                    rank, world, dist, device=dev)

    ecfg = capi.esdf_cfg(min_distance_m=trunc / 2)  # ros_params.h:136-137
    esdf_ms = [0.0]

    def step(i):
        pose, dp, dc = d_frames[i % len(d_frames)]
        n_i = n_pts_all[i % len(d_frames)]
        if sharded is None:
            gm.integrate_device(kind, cfg, pose[0], pose[1], dp.data_ptr(), dc.data_ptr(), n_i)
            if args.esdf:
                tt = gm.timing() if timing_on[0] else None
                cc = gm.counters()
                gm.esdf_update(ecfg, batch=False, clear_updated_flag=True)
                if timing_on[0]:
                    esdf_ms[0] += gm.timing()["total_ms"]
                    esdf_cnt.update({k: esdf_cnt.get(k, 0) + v for k, v in gm.counters().items() if k.startswith("esdf")})
                    last_tsdf[0] = (tt, cc)
            if args.mesh:
                tt = gm.timing() if (timing_on[0] and not args.esdf) else None
                cc = gm.counters() if not args.esdf else None
                tm0 = time.perf_counter()
                midx, moff = gm.mesh_generate(mcfg, True, True, download=False)
                if timing_on[0]:
                    mesh_acc["s"] += time.perf_counter() - tm0
                    mesh_acc["blocks"] += len(midx)
                    mesh_acc["vertices"] += int(moff[-1])
                    mesh_acc["calls"] += 1
                    if not args.esdf:
                        last_tsdf[0] = (tt, cc)
        else:
            sharded.integrate_shard(kind, cfg, pose[0], pose[1], dp, dc, n_i)

    mcfg = capi.mesh_cfg()
    mesh_acc = {"s": 0.0, "blocks": 0, "vertices": 0, "calls": 0}
    timing_on = [False]
    esdf_cnt = {}
    last_tsdf = [None]

    def barrier():
        if sharded is not None:
            sharded.flush()          # the exchange worker must be idle before a collective of ours
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    gm.enable_timing(True)
    timing_on[0] = True
    if sharded is not None:
        sharded.flush()
        for k in sharded.stats:
            sharded.stats[k] = 0
    stage = {}
    counters = {}
    barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, total):
        step(i)
        if sharded is not None and i < total - 1:
            continue   # per-stage figures from the last frame only: no extra calls between frames
        t, c = last_tsdf[0] if ((args.esdf or args.mesh) and last_tsdf[0]) else (gm.timing(), gm.counters())
        rep = args.steps if sharded is not None else 1
        for k, v in t.items():
            stage[k] = stage.get(k, 0.0) + v * rep
        for k, v in c.items():
            counters[k] = counters.get(k, 0) + v * rep
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # Host-mirror cost, outside the timed region: what a caller that keeps a
    # host Layer coherent pays per frame on top of the integration.
    mirror = None
    if sharded is None and args.mirror_frames > 0 and rank == 0:
        timing_on[0] = False
        staging = gm.pinned_voxels(2048)   # page-locked, reused every frame (vbx_host_alloc)
        gm.clear_updated(capi.UPDATE_MAP)
        torch.cuda.synchronize()
        nb = 0
        nbytes = 0
        t_int = 0.0
        t_mir = 0.0
        for j in range(args.mirror_frames):
            ta = time.perf_counter()
            step(total + j)
            torch.cuda.synchronize()
            tb = time.perf_counter()
            upd = gm.blocks_updated(capi.UPDATE_MAP)
            if len(upd) > staging.shape[0]:     # finer voxels touch more blocks: grow the staging (outside the clock)
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
        mirror = {"frames": M, "blocks_per_frame": round(nb / M, 1), "MB_per_frame": round(nbytes / M / 1e6, 3),
                  "integrate_ms": round(t_int / M * 1e3, 4), "mirror_ms": round(t_mir / M * 1e3, 4),
                  "note": "mirror = list updated blocks + AoS pack kernel + one D2H copy into page-locked staging + clear kMap bits"}

    # The non-default fast modes (results differ from the reference on ~1 % of the voxels, see
    # include/vbx_hip.h), measured beside the bit-exact default for reference.
    variants = None
    if sharded is None and rank == 0 and not args.esdf and not args.no_variants and args.integrator in ("fast", "merged") \
            and args.fast_set == 0 and args.merged_order == 0:
        vkw = {"fast": dict(fast_observed_set=1), "merged": dict(merged_bundle_order=1)}[args.integrator]
        vcfg = capi.tsdf_cfg(default_truncation_distance=trunc, **vkw)
        vm = capi.Map(voxel, 16, max_blocks=max_blocks, device=local_rank)
        vm.set_stream(torch.cuda.current_stream().cuda_stream)
        vsteps = max(10, min(args.steps, 40))
        for i in range(args.warmup + vsteps):
            if i == args.warmup:
                torch.cuda.synchronize()
                tv0 = time.perf_counter()
            pose, dp, dc = d_frames[i % len(d_frames)]
            vm.integrate_device(kind, vcfg, pose[0], pose[1], dp.data_ptr(), dc.data_ptr(), n_pts_all[i % len(d_frames)])
        torch.cuda.synchronize()
        tv = time.perf_counter() - tv0
        vpts = sum(n_pts_all[i % len(d_frames)] for i in range(args.warmup, args.warmup + vsteps))
        name = list(vkw.items())[0]
        variants = {"%s=%d" % name: {"value": round(vpts / tv / 1e6, 3), "unit": "Mpoints/s",
                                     "ms_per_step": round(tv / vsteps * 1e3, 4), "steps": vsteps,
                                     "note": "not bit-exact against the reference (exact observed-voxel set / "
                                             "sorted bundle order); the default mode above is"}}
        vm.close()

    K = args.steps
    pts_timed = sum(n_pts_all[i % len(d_frames)] for i in range(args.warmup, total))
    value = world * pts_timed / dt / 1e6
    out = {
        "metric": "Mpoints/s integrated (640x480 frame, %g m voxels) + achieved HBM GB/s" % voxel,
        "value": round(value, 3), "unit": "Mpoints/s", "n_gpus": world, "steps": K,
        "warmup": args.warmup, "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.integrator.capitalize()}TsdfIntegrator, 640x480 synthetic room scan "
                               "stream (BASELINE configs[1]), %g m voxels / 16^3 blocks, trunc %g m" % (voxel, trunc),
                   "points_per_step": n_pts, "voxel_size": voxel, "voxels_per_side": 16,
                   "semantics": "bit-exact vs the 1-thread reference" if (args.merged_order == 0 and args.fast_set == 0)
                                else "fast mode (merged_bundle_order=%d, fast_observed_set=%d)" % (args.merged_order, args.fast_set),
                   "parallelism": ("1 GPU, whole cloud" if world == 1 else
                                   f"{world} sensors, one ray shard per GPU, RCCL reduce-scatter block merge "
                                   "pipelined behind the next frame's integration, map distributed by block owner")},
    }
    if rank == 0:
        # Roofline of the dominant kernel.  For the Fast integrator that is k_fast_sweep, launched
        # ~20 times per frame by the early-termination solver; its "launch" here is one frame's
        # whole sweep sequence, timed with HIP events on the launch stream inside the library
        # (vbx_get_timing solve_ms) and averaged over the K timed frames.  The per-launch average
        # (kernel_ms / launches_per_step) is the number to compare with rocprofv3's avg duration
        # (profiles/r01d_kernel_stats.md).  Algorithmic bytes per frame:
        #   16 B x N_points + 24 B x U (distinct voxels updated), U counted on the device.
        U = counters.get("voxels_touched", 0) / K
        alg_bytes = 16.0 * n_pts + 24.0 * U
        stages = {k: v / K for k, v in stage.items() if k != "total_ms"}
        dom = max(stages, key=stages.get) if stages else "total_ms"
        dom_ms = stages.get(dom, 0.0)
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        kernel_of_stage = {"solve_ms": "k_fast_sweep", "replay_ms": "k_rsort_scatter", "fold_ms": "k_fold", "emit_ms": "k_ray_emit",
                           "prep_ms": "k_prep_points+sort", "alloc_ms": "k_fast_build_lists", "sort_ms": "rocprim onesweep"}
        kname = kernel_of_stage.get(dom, dom)
        launches = {"solve_ms": counters.get("iterations", 0) / K,
                    "replay_ms": 2.0 * counters.get("replay_rounds", 0) / K}.get(dom, 1.0)   # 2 sort passes per round
        launches = max(launches, 1.0)
        # HBM bytes of that kernel per frame from the PMC passes (FETCH_SIZE + WRITE_SIZE, separate
        # runs, profiles/r01d_pmc_hbm_traffic.json); null when no summary for this kernel exists.
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01d_pmc_hbm_traffic.json")))["per_frame_bytes"]
            if args.integrator == "fast" and args.scene == "room" and kname in pmc:
                traffic = int(pmc[kname]["fetch_bytes"] + pmc[kname]["write_bytes"])
        except (OSError, KeyError, ValueError):
            traffic = None
        kdesc = {"k_fast_sweep": "k_fast_sweep (early-termination solver)",
                 "k_rsort_scatter": "k_rsort_scatter (stable radix sort passes of the observed-set replay rounds)"}.get(kname, kname)
        out["roofline"] = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS,
                           "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic,
                           "kernel": kdesc, "launch": "all launches of one frame (stage %s)" % dom,
                           "kernel_ms": round(dom_ms, 4), "launches_per_step": round(launches, 1),
                           "avg_launch_us": round(dom_ms * 1e3 / max(launches, 1.0), 2),
                           "algorithmic_bytes_per_launch": int(alg_bytes),
                           "stage_ms": {k: round(v, 4) for k, v in stages.items()},
                           "device_total_ms": round(stage.get("total_ms", 0.0) / K, 4),
                           "note": "latency/atomic bound irregular path, far below the HBM roofline (SURVEY 8(d)); "
                                   "traffic = PMC FETCH_SIZE+WRITE_SIZE of this kernel per frame"}
        out["counters_per_step"] = {k: round(v / K, 1) for k, v in counters.items()}
        if args.esdf:
            out["esdf"] = {"ms_per_update": round(esdf_ms[0] / K, 4),
                           "counters_per_update": {k: round(v / K, 1) for k, v in esdf_cnt.items()}}
        if args.mesh and mesh_acc["calls"]:
            n = mesh_acc["calls"]
            out["mesh"] = {"ms_per_update": round(mesh_acc["s"] / n * 1e3, 4),
                           "blocks_per_update": round(mesh_acc["blocks"] / n, 1),
                           "vertices_per_update": round(mesh_acc["vertices"] / n, 1),
                           "note": "host wall clock of vbx_mesh_generate (select + count + scan + emit + block table), "
                                   "vertices left device-resident"}
            # what a host MeshLayer consumer pays on top: the same call plus the copy of the arrays
            timing_on[0] = False
            td = 0.0
            for j in range(8):
                pose, dp, dc = d_frames[(total + j) % len(d_frames)]
                gm.integrate_device(kind, cfg, pose[0], pose[1], dp.data_ptr(), dc.data_ptr(),
                                    n_pts_all[(total + j) % len(d_frames)])
                tm0 = time.perf_counter()
                res = gm.mesh_generate(mcfg, True, True, download=True)
                td += time.perf_counter() - tm0
            out["mesh"]["ms_per_update_with_download"] = round(td / 8 * 1e3, 4)
            if world == 1 and not args.no_cpu_baseline:
                out["mesh"]["cpu_reference"] = cpu_mesh_baseline(frames[:12], args.integrator, voxel)
        if mirror:
            out["host_mirror"] = mirror
        if variants:
            out["variants"] = variants
        out["config"]["scene"] = args.scene
        out["config"]["esdf_after_each_frame"] = bool(args.esdf)
        out["config"]["mesh_after_each_frame"] = bool(args.mesh)
        if world == 1 and not args.no_cpu_baseline:
            nf = args.cpu_frames or 40
            out["cpu_baseline"] = cpu_baseline(frames[:min(nf, len(frames))], args.integrator, voxel)
        print(json.dumps(out), flush=True)
    if sharded is not None:
        sharded.close()
        if rank == 0 and os.environ.get("VBX_PIPE_DEBUG"):
            f = max(sharded.stats["frames"], 1)
            print({k: (round(v / f * 1e3, 3) if k != "frames" else v) for k, v in sharded.stats.items()},
                  file=sys.stderr)
    if world > 1 or force_sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
