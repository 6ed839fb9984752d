"""-m gpu: BASELINE configs[4] END TO END AT SIZE — four 640x480 sensors (scenes.room_sensor_frame), 0.02 m
voxels, default FastTsdfIntegrator Config (the reference's lossy ApproxHashSet replayed exactly), ray-bundle
shards integrated into per-step delta maps and merged into the persistent map — against the CPU oracle doing
the same shard + mergeVoxelAIntoVoxelB serially (tests/shard_ref.py; voxel_utils.cc:10-22,
block_inl.h:112-129).  Both host paths of the exchange (voxblox_amd.multi_gpu.PipelinedShardedTsdfMap and
libvbx_shard.so); the layout every bench run uses — four contiguous ray bands per sensor, sixteen bundles per step,
whatever the number of GPUs (multi_gpu.BANDS_PER_SENSOR) — and also whole sensors and halves; and — where the box has
more than one GPU — one RCCL rank per GPU through both paths.

Tolerances (written here): block sets and observed masks equal, distances and weights within 1e-5
(relative for the weights) of the serial float32 merge, colours +-1 LSB (one rounding after the sum where the
reference rounds at every pairwise blend)."""
import os
import sys

import numpy as np
import pytest

from shard_ref import assert_merged_equal, serial_shard_merge
from voxblox_amd import scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VOXEL = 0.02
_cache = {}


def _shards_of_step(step, world, width=640, height=480, f=320.0, bands=None):
    """[rank] -> [(pos, quat, pts, col)] of one time step, dealt out like bench.py deals configs[4]."""
    from voxblox_amd import multi_gpu
    out = []
    for units in multi_gpu.deal_sensor_units(world, bands=bands):
        mine = []
        for s, b, bands in units:
            key = (s, step, width)
            if key not in _cache:  # (the rank-scaling test uses the same poses with fewer pixels)
                _cache[key] = scenes.room_sensor_frame(s, step, f=f, width=width, height=height)
            pose, pts, col = _cache[key]
            lo, hi = multi_gpu.band_of(pts.shape[0], b, bands)
            mine.append((pose[0], pose[1], pts[lo:hi], col[lo:hi]))
        out.append(mine)
    return out


def _reference(oracle, bands, n_steps):
    key = ("ref", bands, n_steps)
    if key not in _cache:
        ocfg = oracle.tsdf_cfg(default_truncation_distance=4 * VOXEL, integrator_threads=1)
        # one delta map per ray shard (the layout bench.py runs): all shards of the step on the ONE rank of this box
        steps = [[[sh for rank in _shards_of_step(k, 1, bands=bands) for sh in rank]] for k in range(n_steps)]
        _cache[key] = serial_shard_merge(oracle, VOXEL, "fast", ocfg, steps, deltas_per_rank=4 * bands)
    return _cache[key]


@pytest.mark.parametrize("bands", [4, 1, 2], ids=["four_bands_per_sensor_the_bench_layout", "whole_sensors", "two_bands_per_sensor"])
@pytest.mark.parametrize("path", ["torch", "native"])
def test_configs4_full_size_step_equals_serial_oracle_shard_merge(oracle, path, bands):
    """Two full time steps (the second merges into a populated persistent map) of configs[4] on this GPU: every
    shard of a step into a delta map of its own, all of them concurrently, exchange with one rank behind the next
    step, owner merge in shard order."""
    import torch
    from voxblox_amd import capi, multi_gpu, shard_native
    n_steps = 2
    dev = torch.device("cuda", 0)
    cfg = capi.tsdf_cfg(default_truncation_distance=4 * VOXEL)
    pm = capi.Map(VOXEL, 16, max_blocks=16384)
    nu = 4 * bands         # shards per step = delta maps per set: every shard a delta map of its own, integrated concurrently
    sets = [[capi.Map(VOXEL, 16, max_blocks=4096) for _ in range(nu)] for _ in range(2)]
    replay_block_rounds = 0
    if path == "torch":
        sm = multi_gpu.PipelinedShardedTsdfMap(multi_gpu.GpuBackend(pm, dev), [[multi_gpu.GpuBackend(d, dev) for d in ds] for ds in sets],
                                               0, 1, device=dev)
    else:
        sm = shard_native.NativeShard(pm, sets[0][0], 0, 1, shard_native.unique_id())   # a real one-rank RCCL communicator
        for d in sets[0][1:] + sets[1]:
            sm.add_delta(d)
        sm.set_pipelined(True)
    for k in range(n_steps):
        shards = [sh for rank in _shards_of_step(k, 1, bands=bands) for sh in rank]
        dsh = [(p, q, torch.from_numpy(pts).to(dev), torch.from_numpy(col).to(dev), pts.shape[0]) for p, q, pts, col in shards]
        if path == "torch":
            sm.integrate_shards(capi.TSDF_FAST, cfg, dsh)
        else:
            sm.begin_step()
            sm.integrate_shards(capi.TSDF_FAST, cfg, [(p, q, dp.data_ptr(), dc.data_ptr(), n) for p, q, dp, dc, n in dsh])
            replay_block_rounds += sum(d.counters()["replay_block_rounds"] for d in sets[k & 1])
            sm.end_step()
    if path == "native":
        sm.wait()
    sm.close()
    torch.cuda.synchronize()
    got = pm.tsdf_dict()
    ref = _reference(oracle, bands, n_steps)
    assert len(ref) > 3000            # thousands of 0.32 m blocks: this is the full-size map
    worst = assert_merged_equal(got, ref)
    if path == "native" and bands == 1:
        assert replay_block_rounds > 0    # the fine-voxel replay of DESIGN 4.4 really ran (whole frames: millions of probes)
    print(f"configs[4] {path}, {bands} bands per sensor: {len(ref)} blocks, max |dd| {worst[0]:.2e}, max rel dw {worst[1]:.2e}")


# ---------------------------------------------------------------------------------------------------------
# one RCCL rank per GPU (skipped on a one-GPU box: RCCL refuses two ranks on one device)
# ---------------------------------------------------------------------------------------------------------
def _rank_worker(rank, world, port, path, out_dir, n_steps, width, height, f, voxel):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    torch.cuda.set_device(rank)
    import torch.distributed as dist
    from voxblox_amd import capi, multi_gpu, shard_native
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    cfg = capi.tsdf_cfg(default_truncation_distance=4 * voxel)
    pm = capi.Map(voxel, 16, max_blocks=4096, device=rank)
    nu = max(1, len(multi_gpu.deal_sensor_units(world)[rank]))
    sets = [[capi.Map(voxel, 16, max_blocks=2048, device=rank) for _ in range(nu)] for _ in range(2)]
    if path == "torch":
        sm = multi_gpu.PipelinedShardedTsdfMap(multi_gpu.GpuBackend(pm, dev), [[multi_gpu.GpuBackend(d, dev) for d in ds] for ds in sets],
                                               rank, world, dist, device=dev)
    else:
        ids = [shard_native.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        sm = shard_native.NativeShard(pm, sets[0][0], rank, world, ids[0], rank)
        for d in sets[0][1:] + sets[1]:
            sm.add_delta(d)
        sm.set_pipelined(True)
    import test_gpu_sensors4_parity as T
    for k in range(n_steps):
        mine = T._shards_of_step(k, world, width, height, f)[rank]
        dsh = [(p, q, torch.from_numpy(pts).to(dev), torch.from_numpy(col).to(dev), pts.shape[0]) for p, q, pts, col in mine]
        if path == "torch":
            sm.integrate_shards(capi.TSDF_FAST, cfg, dsh)
        else:
            sm.begin_step()
            sm.integrate_shards(capi.TSDF_FAST, cfg, [(p, q, dp.data_ptr(), dc.data_ptr(), n) for p, q, dp, dc, n in dsh])
            sm.end_step()
    if path == "native":
        sm.wait()
    # self-diagnosis for the first box with several GPUs (the driver's scaling run is the first time RCCL runs with more than
    # one rank): what every rank saw of the communicator and what it moved, gathered through the SAME RCCL group and printed once
    st = sm.stats() if path == "native" else dict(sm.stats)
    mine = {"rank": rank, "world_seen_by_torch": dist.get_world_size(), "backend": dist.get_backend(),
            "device": torch.cuda.get_device_name(rank), "device_index": rank, "path": path,
            "steps": int(st.get("steps", st.get("frames", 0))), "sent_blocks": int(st.get("sent_blocks", 0)),
            "received_blocks": int(st.get("received_blocks", 0)), "payload_bytes": int(st.get("payload_bytes", 0))}
    # an all-reduce over the device tensors: proves the RCCL ring itself spans `world` ranks (sum of ranks = world (world - 1) / 2)
    t = torch.tensor([float(rank)], device=dev)
    dist.all_reduce(t)
    mine["rccl_allreduce_sum_of_ranks"] = float(t.item())
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    if rank == 0:
        print(f"[multi-GPU diagnosis] path={path} world={world} RCCL all-reduce of ranks = {mine['rccl_allreduce_sum_of_ranks']} "
              f"(expected {world * (world - 1) / 2})", flush=True)
        for e in everyone:
            print("[multi-GPU diagnosis] rank %(rank)d on %(device)s (cuda:%(device_index)d): %(steps)d steps, sent %(sent_blocks)d blocks / "
                  "%(payload_bytes)d B, received %(received_blocks)d rows as owner" % e, flush=True)
        import json
        scratch = os.path.join(ROOT, "gpurun_out")
        with open(os.path.join(scratch if os.path.isdir(scratch) else out_dir, f"multi_gpu_diagnosis_{path}.json"), "w") as fh:
            json.dump(everyone, fh, indent=1)
    assert mine["rccl_allreduce_sum_of_ranks"] == world * (world - 1) / 2
    sm.close()
    torch.cuda.synchronize()
    owned = pm.tsdf_dict()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"),
             keys=np.array(list(owned.keys()), np.int32).reshape(-1, 3),
             d=np.stack([v[0] for v in owned.values()]) if owned else np.zeros((0, 4096), np.float32),
             w=np.stack([v[1] for v in owned.values()]) if owned else np.zeros((0, 4096), np.float32),
             c=np.stack([v[2] for v in owned.values()]) if owned else np.zeros((0, 4096, 4), np.uint8))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("path", ["torch", "native"])
def test_one_rccl_rank_per_gpu_equals_serial_oracle_merge(oracle, path, tmp_path):
    """world = min(#GPUs, 8) real RCCL ranks: shards dealt out as bench.py deals them at that world size, the sparse
    all-to-all-v between ranks, owner merge; the union of the ranks' owned blocks == the serial oracle merge of
    the same deltas (added in rank order).  Smaller frames (160x120, 0.05 m): the exchange is what is under test."""
    import torch
    import torch.multiprocessing as mp
    from test_multi_gpu_gloo import _free_port
    from voxblox_amd import multi_gpu
    n_gpu = torch.cuda.device_count()
    if n_gpu < 2:
        pytest.skip("one GPU on this box: RCCL refuses two ranks on one device (the world-2 protocol runs under gloo in "
                    "tests/test_multi_gpu_gloo.py and test_gpu_multi_merge.py)")
    world = min(n_gpu, 8)
    n_steps, width, height, f, voxel = 3, 160, 120, 80.0, 0.05
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_rank_worker, args=(r, world, port, path, str(tmp_path), n_steps, width, height, f, voxel))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
        assert p.exitcode == 0
    merged = {}
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        if z["keys"].shape[0]:
            assert np.all(multi_gpu.owner_of(z["keys"], world) == r)      # the map is distributed by block owner
        for i, k in enumerate(z["keys"]):
            kk = tuple(int(v) for v in k)
            assert kk not in merged
            merged[kk] = (z["d"][i], z["w"][i], z["c"][i], 7)
    ocfg = oracle.tsdf_cfg(default_truncation_distance=4 * voxel, integrator_threads=1)
    steps = [_shards_of_step(k, world, width, height, f) for k in range(n_steps)]
    ref = serial_shard_merge(oracle, voxel, "fast", ocfg, steps, deltas_per_rank=16)   # every shard a delta map of its own
    assert len(ref) > 100
    assert_merged_equal(merged, ref)
