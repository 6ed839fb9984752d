"""What ray-bundle shard + merge does to the map, against the reference integrating the same clouds into one map
(round-3 verdict: "nobody has measured how far the result is from integratePointCloud x 4").  CPU only: both sides are
the oracle; the shard + merge side is the serial form the GPU paths are tested against (tests/shard_ref.py).  The
numbers at BASELINE size live in profiles/r04_shard_divergence_*.json (tools/shard_divergence.py); this test pins the
two facts the default follows from on a small case."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_whole_sensor_bundles_equal_the_reference_and_row_bands_do_not(oracle):
    import shard_divergence as sd
    out = sd.measure(0.1, 3, 160, 120)
    first, last = out["after_step"]["1"], out["after_step"]["3"]
    whole = "4 bundles (whole sensors), apply_caps off"
    bands = "16 bundles (four row bands per sensor), apply_caps off"
    # one bundle per sensor cloud: every cloud meets the lossy sets of FastTsdfIntegrator exactly as in the reference, the
    # weighted sums commute with its running average — the first step is the reference's map, later steps stay within 2e-2 m
    # of it on a few per cent of the voxels (the reference clamps to +-trunc after EVERY cloud, the merge once per step)
    assert first[whole]["blocks_only_merged"] == 0 and first[whole]["blocks_only_reference"] <= 8   # (blocks the reference allocated without writing a voxel)
    assert first[whole]["observed_mask_differences"] == 0 and first[whole]["frac_gt_1e-4_m"] == 0.0
    assert last[whole]["observed_mask_differences"] == 0 and last[whole]["frac_gt_1e-4_m"] < 0.1 and last[whole]["max_m"] < 0.05
    assert abs(last[whole]["weight_ratio"] - 1.0) < 1e-3
    # four row bands per sensor: every band starts with empty ApproxHashSets, so rays the reference drops are integrated —
    # another map (that is the price of the sixteen-bundle form, stated next to its throughput)
    assert first[bands]["frac_gt_1e-4_m"] > 0.05 and first[bands]["observed_mask_differences"] > 0
