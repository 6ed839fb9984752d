"""bench.py's driver-facing line: ONE JSON object, under 4 KB, carrying `roofline` and `cpu_baseline`.

Round 4's line had grown to 21 KB and the driver could not parse it (BENCH_r04.parsed = null); the full result now goes
to bench_detail.json and the line is built by bench.compact_line.  CPU test: the builder on round 4's full result and on an
inflated one.  GPU test: the driver's own command shape end to end.
"""
import copy
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _check_line(text):
    assert "\n" not in text
    assert len(text) < 4096, len(text)
    j = json.loads(text)
    for k in REQUIRED:
        assert k in j, k
    assert j["config"]["workload"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in j["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in j["cpu_baseline"], k
    return j


def test_compact_line_of_round_4s_full_result_fits_and_keeps_the_contract():
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))
    assert len(json.dumps(full)) > 15000          # the line the driver could not parse
    j = _check_line(bench.compact_line(full, "bench_detail.json"))
    assert j["value"] == full["value"] and j["ms_per_step"] == full["ms_per_step"]
    assert j["roofline"]["frac"] == full["roofline"]["frac"]
    assert set(j["cpu_baseline"]["by_threads"]) == set(full["cpu_baseline"]["by_threads"])
    assert len(j["legs"]) == len(full["other_configs"])
    for leg in j["legs"].values():
        assert "value" in leg and "ms_per_step" in leg


def test_compact_line_sheds_optional_parts_instead_of_overflowing():
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))
    fat = copy.deepcopy(full)
    fat["other_configs"] = {"leg %d %s" % (i, "x" * 60): copy.deepcopy(v) for i in range(6) for v in full["other_configs"].values()}
    fat["cpu_baseline"]["by_threads"] = {str(t): {"value": 1.0 + t} for t in range(1, 120)}
    j = _check_line(bench.compact_line(fat, "bench_detail.json"))
    assert j["value"] == full["value"]


def _pmc_ratio(roof):
    """counter bytes per STEP of the line's dominant kernel, read from the PMC file the line names, over the line's
    algorithmic bytes per step"""
    pmc = json.load(open(os.path.join(ROOT, roof["traffic_source"])))
    k = pmc["per_frame_bytes"][roof["kernel"]]
    return (k["fetch_bytes"] + k["write_bytes"]) / roof["algorithmic_bytes_per_step"], pmc["total_bytes_per_frame"] / roof["algorithmic_bytes_per_step"]


def test_roofline_traffic_is_counter_bytes_per_step_of_the_dominant_kernel():
    """Round 5's line put bytes per LAUNCH in `traffic` next to algorithmic bytes per STEP (12.7 MB against 10.1 MB looked like
    1.25x where the kernel moves 25x and the step 88x the compulsory bytes).  `traffic` is per step now: its ratio to
    algorithmic_bytes_per_step equals the PMC file's, for the dominant kernel and for all kernels."""
    import bench
    rows = [{"kernel": "k_fast_sweep", "us_per_step": 469.3, "launches_per_step": 20.2, "avg_us": 23.23},
            {"kernel": "k_rsort_fused", "us_per_step": 380.0, "launches_per_step": 19.6, "avg_us": 19.4}]
    alg = 10143285
    r = bench.roofline_from(rows, alg, 1.4226, "16 B x points + 24 B x distinct voxels updated per frame")
    assert r["kernel"] == "k_fast_sweep" and r["traffic"] is not None
    short = bench._short_roofline(r)
    for roof in (r, short):
        roof = dict(roof, traffic_source=(roof.get("traffic_source") or roof["traffic_detail"]["source"]))
        dom, allk = _pmc_ratio(roof)
        assert abs(roof["traffic"] / roof["algorithmic_bytes_per_step"] - dom) <= 0.01 * dom
        assert abs(roof["traffic_all_kernels_per_step"] / roof["algorithmic_bytes_per_step"] - allk) <= 0.01 * allk
        assert abs(roof["traffic_over_algorithmic"] - dom) <= 0.01 * dom + 0.01
    assert abs(r["traffic"] - r["traffic_per_launch"] * 20.2) <= 0.01 * r["traffic"]


def test_host_cpu_info_names_affinity_and_quota():
    import bench
    i = bench.host_cpu_info()
    assert i["host_hw_threads"] >= 1 and "affinity_threads" in i and "cgroup_cpu_quota_cores" in i
    assert 1 <= bench._usable_cores() <= i["host_hw_threads"]


@pytest.mark.gpu
def test_the_drivers_command_prints_one_parseable_line_under_4_kb(tmp_path):
    """`python bench.py --gpus 1 --steps K --warmup W` as the driver runs it (short K so that the test stays short; the
    secondary legs shorten themselves with it): the LAST stdout line is the result."""
    detail = tmp_path / "detail.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
                        "--cpu-threads", "1,16", "--detail-out", str(detail)], capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    j = _check_line(lines[-1])
    assert j["n_gpus"] == 1 and j["steps"] == 4 and j["warmup"] == 2
    assert j["value"] > 0 and j["roofline"]["frac"] > 0
    legs = j["legs"]
    assert len(legs) == 3 and all("value" in v for v in legs.values()), legs
    esdf = [v for k, v in legs.items() if "configs[3]" in k][0]["esdf"]
    assert esdf["voxels_differing_from_reference"] == 0 and esdf["voxels_compared"] > 100000, esdf
    sens = [v for k, v in legs.items() if "configs[4]" in k][0]
    assert sens["ray_bundles_per_step"] == 4
    full = json.load(open(detail))
    assert full["value"] == j["value"] and "kernels" in full
    # the line says what it measures: which entry point `value` is, the same for the drop-in figure, counter traffic per step
    assert "vbx_tsdf_integrate_device" in j["config"]["entry"]
    assert "integratePointCloud" in j["dropin_path"]["entry"]
    assert len(j["dropin_path"]["by_host_blocks"]) == 3, j["dropin_path"]
    for hb, v in j["dropin_path"]["by_host_blocks"].items():
        assert isinstance(v, list) and v[0] > 0 and v[1] is not None and v[1] < 1.0, (hb, v)   # [Mpoints/s, reconcile ms]
    roof = j["roofline"]
    if roof.get("traffic") is not None:
        dom, allk = _pmc_ratio(roof)
        assert abs(roof["traffic"] / roof["algorithmic_bytes_per_step"] - dom) <= 0.01 * dom
        assert abs(roof["traffic_all_kernels_per_step"] / roof["algorithmic_bytes_per_step"] - allk) <= 0.01 * allk
    assert "affinity_threads" in j["cpu_baseline"] and "cgroup_cpu_quota_cores" in j["cpu_baseline"]


def test_compact_line_of_a_multi_rank_result_and_of_a_failed_leg():
    """The N > 1 shape (weak scaling, an `exchange` block, configs[4] as the only leg) and a leg that failed: the line keeps
    the contract's keys, carries the leg's error text in short form and never the traceback."""
    import bench
    out = {"metric": "Mpoints/s integrated (640x480 frame, 0.05 m voxels) + achieved HBM GB/s", "unit": "Mpoints/s", "n_gpus": 8,
           "steps": 20, "warmup": 5, "higher_is_better": True, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "value": 1500.0, "ms_per_step": 1.64, "scaling": "weak",
           "config": {"workload": "BASELINE configs[1] on every GPU: " + "x" * 400, "points_per_step": 8 * 307200,
                      "points_per_step_per_gpu": 307200, "voxel_size": 0.05, "voxels_per_side": 16, "world_size_seen": 8,
                      "semantics": "y" * 500, "parallelism": "z" * 300},
           "exchange": {"payload_bytes_per_step": 6_000_000, "exchange_ms_per_step": 0.4, "integrate_ms_per_step": 1.5,
                        "wait_ms_per_step": 0.0, "note": "n" * 300},
           "other_configs": {"configs[4]": {"error": "Traceback (most recent call last):\n" + "frame\n" * 200}}}
    text = bench.compact_line(out, "bench_detail.json")
    assert len(text) < 4096 and "\n" not in text
    j = json.loads(text)
    assert j["n_gpus"] == 8 and j["scaling"] == "weak" and j["value"] == 1500.0
    assert j["roofline"] is None and j["cpu_baseline"] is None          # ranks > 1 carry neither (rank 0 at N = 1 only)
    assert j["exchange"]["payload_bytes_per_step"] == 6_000_000
    assert len(j["config"]["workload"]) <= 160 and "semantics" not in j["config"]
    leg = j["legs"]["configs[4]"]
    assert set(leg) == {"error"} and len(leg["error"]) <= 120
