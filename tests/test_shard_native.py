"""libvbx_shard.so (include/vbx_shard.h): the C++ / RCCL host path of the multi-GPU sharding.
CPU: the library loads, exports every declared symbol, and its owner function is the one the Python /
gloo-tested protocol uses.  -m gpu: one rank, with and without a real one-rank RCCL communicator, produces
exactly the map voxblox_amd.multi_gpu.ShardedTsdfMap produces from the same shards."""
import ctypes
import re
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_library_exports_every_declared_symbol():
    from voxblox_amd import shard_native
    hdr = open(os.path.join(ROOT, "include", "vbx_shard.h")).read()
    declared = set(re.findall(r"\b(vbx_shard_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(shard_native.EXPORTED_SYMBOLS), declared ^ set(shard_native.EXPORTED_SYMBOLS)
    L = ctypes.CDLL(shard_native.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name


def test_owner_function_matches_python_protocol():
    from voxblox_amd import multi_gpu, shard_native
    rng = np.random.RandomState(1)
    keys = rng.randint(-3000, 3000, (500, 3)).astype(np.int32)
    for world in (1, 2, 3, 8):
        want = multi_gpu.owner_of(keys, world)
        got = np.array([shard_native.owner_of(k, world) for k in keys])
        assert np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("with_rccl", [False, True])
def test_native_shard_equals_python_shard(with_rccl):
    import torch
    from voxblox_amd import capi, multi_gpu, scenes, shard_native
    voxel = 0.1
    cfg = capi.tsdf_cfg(default_truncation_distance=4 * voxel)
    frames = [scenes.room_frame(6 * k, 100, f=40.0, width=80, height=60) for k in range(4)]
    dev = torch.device("cuda", 0)
    ref = multi_gpu.ShardedTsdfMap(multi_gpu.GpuBackend(capi.Map(voxel, 16, max_blocks=2048), dev),
                                   multi_gpu.GpuBackend(capi.Map(voxel, 16, max_blocks=2048), dev), 0, 1)
    pm, dm = capi.Map(voxel, 16, max_blocks=2048), capi.Map(voxel, 16, max_blocks=2048)
    ns = shard_native.NativeShard(pm, dm, 0, 1, shard_native.unique_id() if with_rccl else None)
    for pose, pts, col in frames:
        n = pts.shape[0]
        dp, dc = torch.from_numpy(pts).to(dev), torch.from_numpy(col).to(dev)
        shards = [(0, n // 2), (n // 2, n)]          # two ray bands of the frame in one step
        ref.integrate_shards(capi.TSDF_FAST, cfg, [(pose[0], pose[1], dp[a:b], dc[a:b], b - a) for a, b in shards])
        ns.begin_step()
        for a, b in shards:
            ns.integrate(capi.TSDF_FAST, cfg, pose[0], pose[1], dp[a:b].data_ptr(), dc[a:b].data_ptr(), b - a)
        ns.end_step()
    torch.cuda.synchronize()
    a, b = ref.p.m.tsdf_dict(), pm.tsdf_dict()
    assert set(a) == set(b) and len(a) > 20
    for k in a:
        assert np.array_equal(a[k][0].view(np.uint32), b[k][0].view(np.uint32))
        assert np.array_equal(a[k][1].view(np.uint32), b[k][1].view(np.uint32))
        assert np.array_equal(a[k][2], b[k][2]) and a[k][3] == b[k][3]
    st = ns.stats()
    assert st["steps"] == len(frames) and st["sent_blocks"] == st["received_blocks"] > 0
    assert st["payload_bytes"] == st["sent_blocks"] * 3 * 4096 * 4
    ns.close()
