"""The device code of the reference-order ESDF replay (voxblox_amd/csrc/vbx_esdf_replay_core.hpp), run WITHOUT a GPU.

The replay is written as phases — plain functions of (arguments, thread id) with no synchronisation inside — plus a
control function that picks the next phase.  tools/esdf_order_model.cc compiles that very header for the host (the HIP
builtins it uses become one-thread functions), runs every phase as a loop over its thread ids in a shuffled order on
flat copies of the oracle's layer and queue, and compares the layer it ends with against the oracle's sequential
processRaiseSet / processOpenSet voxel by voxel (distance bits, flags, parent).  Checker: oracle/ (test infrastructure).
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "esdf_order_model.cc")
LIB = os.path.join(ROOT, "tools", "libesdf_order_model.so")
CORE = os.path.join(ROOT, "voxblox_amd", "csrc", "vbx_esdf_replay_core.hpp")


@pytest.fixture(scope="module")
def model():
    deps = [SRC, CORE] + [os.path.join(ROOT, "oracle", f) for f in ("vbx_esdf.hpp", "vbx_tsdf.hpp", "vbx_core.hpp")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-I", os.path.join(ROOT, "oracle"),
                               SRC, "-o", LIB])
    L = C.CDLL(LIB)
    fp = C.POINTER(C.c_float)
    L.eom_create.restype = C.c_void_p
    L.eom_create.argtypes = [C.c_float]
    L.eom_integrate.argtypes = [C.c_void_p, fp, fp, fp, C.POINTER(C.c_uint8), C.c_size_t]
    L.eom_update.argtypes = [C.c_void_p, C.c_int]
    L.eom_set_mode.argtypes = [C.c_void_p, C.c_int]
    L.eom_update_parallel.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
    L.eom_update_parallel.restype = C.c_long
    L.eom_robot.argtypes = [C.c_void_p, fp]
    L.eom_check_counts.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    return L


def _run(L, n_frames, voxel, sub, kmax, smax, max_iters, env=None, robot=False, checks=None):
    sys.path.insert(0, ROOT)
    from voxblox_amd import scenes
    fp = C.POINTER(C.c_float)
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    os.environ["EOM_THREADS"] = "1"
    try:
        h = L.eom_create(voxel)
        L.eom_set_mode(h, 2)          # the second ESDF layer is updated by the emulated device code
        diffs = []
        for i in range(n_frames):
            pose, pts, col = scenes.room_frame(i, 100)
            pos = np.ascontiguousarray(pose[0], np.float32)
            q = np.ascontiguousarray(pose[1], np.float32)
            pts = np.ascontiguousarray(pts[::sub], np.float32)
            col = np.ascontiguousarray(col[::sub], np.uint8)
            L.eom_integrate(h, pos.ctypes.data_as(fp), q.ctypes.data_as(fp), pts.ctypes.data_as(fp),
                            col.ctypes.data_as(C.POINTER(C.c_uint8)), pts.shape[0])
            if robot:                 # EsdfIntegrator::addNewRobotPosition on both layers before the update
                L.eom_robot(h, pos.ctypes.data_as(fp))
            L.eom_update(h, 0)        # the oracle's sequential update on the first ESDF layer
            diffs.append(L.eom_update_parallel(h, kmax, smax, max_iters))
            if checks is not None:    # EOM_CHECK=1: fixed points refolded in full / inconsistent ones
                out = (C.c_ulonglong * 2)()
                L.eom_check_counts(h, out)
                checks.append((int(out[0]), int(out[1])))
        return diffs
    finally:
        os.environ.pop("EOM_THREADS", None)
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_emulated_replay_equals_the_sequential_wavefront(model):
    """Three incremental updates (raise_ and open_ both busy from the second on): 0 voxels differ."""
    assert _run(model, 3, 0.1, 64, 8192, 256, 64) == [0, 0, 0]


def test_emulated_replay_with_every_early_stop(model):
    """Tiny capacities and caps: 2 048 base records per super-step, excursions cut at 64 records, 8 iterations per
    super-step, 2 500 records / targets in all — iteration caps, excursion cuts, event-list overflow and record / target
    exhaustion all fire, and the committed prefixes still add up to the reference's result."""
    assert _run(model, 3, 0.1, 64, 2048, 64, 8, env={"EOM_REC_CAP": "2500", "EOM_TGT_CAP": "2500"}) == [0, 0, 0]


def test_emulated_replay_without_the_offer_filter(model):
    """Every offer an event (Cfg::filter = 0): the unfiltered form is the definition the filter has to agree with."""
    assert _run(model, 2, 0.1, 64, 8192, 256, 64, env={"EOM_FILTER_LEVEL": "0"}) == [0, 0]


@pytest.mark.parametrize("env", [{}, {"EOM_MULTI_QUEUE": "1", "EOM_BUCKETS": "5"}])
def test_emulated_replay_starts_from_queues_that_hold_robot_sphere_entries(model, env):
    """addNewRobotPosition before every update (esdf_server.cc:219-230): raise_ and open_ are not empty when the voxel walk
    starts, open_ holds voxels whose in_queue flag is clear (esdf_integrator.cc:84 does not set it) and, from the second
    frame on, blocks are listed twice.  With multi_queue and five buckets a bucket holds the same neighbourhoods many
    times over: the case in which the first record of a super-step may find a target's event list full and the
    super-step is retried with fewer records (rp_retry_smaller)."""
    e = dict(env, EOM_SPHERES="0.6,1.5")
    assert _run(model, 3, 0.1, 16, 8192, 256, 64, env=e, robot=True) == [0, 0, 0]


@pytest.mark.parametrize("caps", [False, True])
def test_every_fixed_point_is_consistent_for_all_targets(model, caps):
    """The iteration folds only the targets somebody marked dirty.  EOM_CHECK folds ALL targets once more at every fixed
    point: nothing in front of the cut may change, i.e. no target was left with a stale order of its events (rankings
    mark the targets of the records they move: rp_mark_rec_targets).  Also with the capacities that force every early
    stop."""
    checks = []
    env = {"EOM_CHECK": "1"}
    if caps:
        env.update(EOM_REC_CAP="2500", EOM_TGT_CAP="2500")
        d = _run(model, 3, 0.1, 64, 2048, 64, 8, env=env, checks=checks)
    else:
        d = _run(model, 3, 0.1, 16, 8192, 256, 64, env=env, checks=checks)
    assert d == [0, 0, 0]
    assert sum(r for r, _ in checks) > 20
    assert [f for _, f in checks] == [0, 0, 0]


@pytest.mark.parametrize("env", [{"EOM_EV": "128"}, {"EOM_EV": "64"}, {"EOM_MARK_MOVED": "1"}, {"EOM_NO_MARK_MOVED": "1"},
                                 {"EOM_NO_FOLD_ALL": "1"}, {"EOM_NO_TGT_CLAIM": "1"}, {"EOM_NO_SLOT_BY_BASE": "1"}, {"EOM_MEMBER_LIMIT": "1", "EOM_SMAX_SMALL": "1"}])
def test_emulated_replay_with_every_switch_the_other_way(model, env):
    env = dict(env)
    """rp::Cfg's switches (events per target, which targets a ranking marks, the dirty list of PH_PLACE_BASE, claimed
    targets, the ranking lists' slots) change how the replay gets there, never where: 0 voxels differ with each of them set the other way."""
    # (EOM_MEMBER_LIMIT: the serial ranking stops where the device's has to — behind a record with a child outside the member
    # list —; excursions cut at 32 records so that lists do fill up at this size)
    smax = 32 if env.pop("EOM_SMAX_SMALL", None) else 256
    assert _run(model, 2, 0.1, 32, 8192, smax, 64, env=env) == [0, 0]
