"""How far is the map of ray-bundle shard + merge (voxblox_amd.multi_gpu / libvbx_shard.so) from the map the
reference makes of the same clouds?  CPU only (oracle = the checker; no GPU code runs here).

Per step the four sensors of BASELINE configs[4] produce one cloud each.
  reference   one persistent map, ONE FastTsdfIntegrator, integratePointCloud x 4 per step in sensor order
              (per-call ApproxHashSet reset, per-update clamp to +-trunc and weight cap, tsdf_integrator.cc:205-208)
  shard+merge every bundle into a zeroed delta map by a fresh integrator, the deltas' weighted sums added per block, merged
              into the persistent voxel with mergeVoxelAIntoVoxelB (voxel_utils.cc:10-22: no clamp, no cap) —
              tests/shard_ref.py, the serial form the GPU paths are tested against; bundles = whole sensors (4 per step)
              or four row bands per sensor (16 per step); with apply_caps the merged voxel is clamped / capped afterwards.
Reported after step 1 and after the last step: blocks only one side has, fraction of commonly observed voxels whose
distances differ by more than 1e-4 m, max / rmse, ratio of the summed weights, largest weight.

usage: shard_divergence.py [VOXEL=0.05] [STEPS=10] [WIDTH=640] [HEIGHT=480]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py as O  # noqa: E402
from voxblox_amd import scenes  # noqa: E402
from test_multi_gpu_gloo import merge_A_into_B  # noqa: E402


def compare(got, ref):
    only_g = len(set(got) - set(ref))
    only_r = len(set(ref) - set(got))
    n = n4 = 0
    se = 0.0
    worst = 0.0
    wg = wr = 0.0
    wmax = 0.0
    mask_diff = 0
    for k in set(got) & set(ref):
        gd, gw = got[k][0], got[k][1]
        rd, rw = ref[k][0], ref[k][1]
        both = (gw > 0) & (rw > 0)
        mask_diff += int(((gw > 0) != (rw > 0)).sum())
        d = np.abs(gd[both] - rd[both]).astype(np.float64)
        n += int(both.sum())
        n4 += int((d > 1e-4).sum())
        se += float((d ** 2).sum())
        if d.size:
            worst = max(worst, float(d.max()))
        wg += float(gw[both].sum())
        wr += float(rw[both].sum())
        wmax = max(wmax, float(gw.max()))
    return {"blocks_only_merged": only_g, "blocks_only_reference": only_r, "observed_mask_differences": mask_diff,
            "common_voxels": n, "frac_gt_1e-4_m": round(n4 / max(n, 1), 5), "max_m": round(worst, 5),
            "rmse_m": round((se / max(n, 1)) ** 0.5, 6), "weight_ratio": round(wg / max(wr, 1e-30), 4), "max_weight_merged": round(wmax, 1)}


def measure(voxel, steps, width, height, verbose=False):
    trunc = 4 * voxel
    cfg = O.tsdf_cfg(default_truncation_distance=trunc, integrator_threads=1)
    max_weight = 10000.0
    L = O.lib()
    ref = O.OracleMap(voxel, 16)
    L.orc_fast_reset_counter_set(0)
    ref_it = ref.tsdf_integrator("fast", cfg)
    variants = {"4 bundles (whole sensors)": 1, "16 bundles (four row bands per sensor)": 4}
    merged = {(name, caps): {} for name in variants for caps in (False, True)}
    out = {"voxel": voxel, "steps": steps, "resolution": [width, height], "after_step": {}}
    nv = 16 ** 3
    for st in range(steps):
        clouds = [scenes.room_sensor_frame(s, st, 25, width=width, height=height, f=320.0 * width / 640.0) for s in range(4)]
        for pose, pts, col in clouds:
            L.orc_fast_reset_counter_set(0)
            ref_it.integrate(pose[0], pose[1], pts, col)
        for name, bands in variants.items():
            sums = {}
            for pose, pts, col in clouds:
                rows = height // bands
                for b in range(bands):
                    sl = slice(b * rows * width, (b + 1) * rows * width if b < bands - 1 else None)
                    m = O.OracleMap(voxel, 16)
                    L.orc_fast_reset_counter_set(0)
                    m.tsdf_integrator("fast", cfg).integrate(pose[0], pose[1], pts[sl], col[sl])
                    for key, (d, w, c, _) in m.tsdf_dict().items():
                        sA = np.stack([w * d, w] + [w * c[:, ch].astype(np.float32) for ch in range(4)]).astype(np.float32)
                        sums[key] = (sums[key] + sA).astype(np.float32) if key in sums else sA
                    del m
            for caps in (False, True):
                tgt = merged[(name, caps)]
                for key, sA in sums.items():
                    if not (sA[1] > 0).any():
                        continue
                    dB, wB, cB = tgt.get(key, (np.zeros(nv, np.float32), np.zeros(nv, np.float32), np.zeros((nv, 4), np.uint8)))
                    d, w, c = merge_A_into_B(sA, dB, wB, cB)
                    if caps:   # what the reference's updateTsdfVoxel does after every update (tsdf_integrator.cc:205-208)
                        d = np.clip(d, -trunc, trunc).astype(np.float32)
                        w = np.minimum(w, max_weight).astype(np.float32)
                    tgt[key] = (d, w, c)
        if st == 0 or st == steps - 1:
            r = ref.tsdf_dict()
            out["after_step"][str(st + 1)] = {f"{name}, apply_caps {'on' if caps else 'off'}": compare(merged[(name, caps)], r)
                                               for name in variants for caps in (False, True)}
        if verbose:
            print("step", st + 1, "done", file=sys.stderr, flush=True)
    return out


if __name__ == "__main__":
    a = sys.argv
    print(json.dumps(measure(float(a[1]) if len(a) > 1 else 0.05, int(a[2]) if len(a) > 2 else 10, int(a[3]) if len(a) > 3 else 640,
                             int(a[4]) if len(a) > 4 else 480, verbose=True), indent=1))
