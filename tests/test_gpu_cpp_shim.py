"""-m gpu: the C++ host shim (voxblox_amd/host/vbx_integrators.hpp — reference class names
and signatures over the C-ABI) driven like test/test_sdf_integrators.cc, checked against the
oracle on the same cloud."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shim_demo_matches_oracle(oracle):
    exe = os.path.join(ROOT, "tests", "cpp", "shim_demo")
    if not os.path.exists(exe):
        import __graft_entry__ as g
        g.build()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "shim OK" in out.stdout
    sh = re.search(r"sharded blocks=(\d+) steps=2 sent=(\d+) received=(\d+)", out.stdout)
    assert sh and int(sh.group(1)) > 0 and int(sh.group(2)) == int(sh.group(3)), out.stdout   # vbx_sharded.hpp over RCCL
    got = {}
    for line in out.stdout.splitlines():
        m = re.match(r"(simple|merged|fast) blocks=(\d+) observed=(\d+) sum_w=([-\d.]+) sum_d=([-\d.]+)", line)
        if m:
            got[m.group(1)] = (int(m.group(2)), int(m.group(3)), float(m.group(4)), float(m.group(5)))
    assert set(got) == {"simple", "merged", "fast"}
    mesh_line = re.search(r"mesh blocks=(\d+) vertices=(\d+) mean_z=([-\d.]+)", out.stdout)
    assert mesh_line, out.stdout
    # same cloud as tests/cpp/shim_demo.cc
    u, v = np.meshgrid(np.arange(64), np.arange(48))
    x = (u.reshape(-1) + 0.5 - 32) / 32.0
    y = (v.reshape(-1) + 0.5 - 24) / 32.0
    pts = np.stack([(3.0 * x).astype(np.float32), (3.0 * y).astype(np.float32),
                    np.full(x.shape, 3.0, np.float32)], 1)
    col = np.stack([u.reshape(-1), v.reshape(-1), np.full(x.shape, 40), np.full(x.shape, 255)], 1).astype(np.uint8)
    pos = np.zeros(3, np.float32); q = np.array([1, 0, 0, 0], np.float32)
    for kind, kw in (("simple", {}), ("merged", {}),   # Merged: the reference's own bundle order (default)
                     ("fast", {})):                             # Fast: the reference's own ApproxHashSet (default)
        oracle.lib().orc_fast_reset_counter_set(0)
        m = oracle.OracleMap(0.1, 16)
        it = m.tsdf_integrator(kind, oracle.tsdf_cfg(default_truncation_distance=np.float32(0.4),
                                                     integrator_threads=1, **kw))
        it.integrate(pos, q, pts, col)
        d = m.tsdf_dict()
        obs = sum(int((w > 1e-6).sum()) for _, w, _, _ in d.values())
        sw = sum(float(w[w > 1e-6].astype(np.float64).sum()) for _, w, _, _ in d.values())
        sd = sum(float(dd[w > 1e-6].astype(np.float64).sum()) for dd, w, _, _ in d.values())
        blocks, gobs, gsw, gsd = got[kind]
        assert blocks == len(d) and gobs == obs, (kind, got[kind], len(d), obs)
        assert abs(gsw - sw) <= 1e-5 * max(1.0, abs(sw)) and abs(gsd - sd) <= 1e-5 * max(1.0, abs(sd)), kind

    # the mesh the shim's MeshIntegrator stored for the Fast layer == the oracle's MeshIntegrator
    ml = m.mesh_layer()
    ml.generate(True, True)
    meshes = ml.as_dict()
    n_vert = sum(v["vertices"].shape[0] for v in meshes.values())
    mean_z = sum(float(v["vertices"][:, 2].astype(np.float64).sum()) for v in meshes.values()) / n_vert
    assert int(mesh_line.group(1)) == len(meshes) and int(mesh_line.group(2)) == n_vert
    assert abs(float(mesh_line.group(3)) - mean_z) < 1e-5
