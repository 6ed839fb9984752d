#!/bin/bash
# round 6, last session: the iteration cap against the FIRST update of a map and the batch update (VBX_RP_MAX_ITERS sweep)
OUT=gpurun_out/${1:-r06cap}; mkdir -p $OUT
export TMPDIR=/tmp
cat > /tmp/cap_probe.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
torch.cuda.init()
from voxblox_amd import capi, scenes
dev = torch.device("cuda", 0)
frames = [scenes.room_frame(k, 100) for k in range(6)]
d = [(p, torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)) for p, a, c in frames]
for rep in range(2):
    gm = capi.Map(0.05, 16, max_blocks=8192)
    cfg = capi.tsdf_cfg(default_truncation_distance=0.2)
    ecfg = capi.esdf_cfg(min_distance_m=0.1, reference_order=1)
    ts = []
    for i, (pose, dp, dc) in enumerate(d):
        gm.integrate_device(capi.TSDF_FAST, cfg, pose[0], pose[1], dp.data_ptr(), dc.data_ptr(), dp.shape[0])
        t0 = time.perf_counter(); gm.esdf_update(ecfg, batch=False, clear_updated_flag=True); ts.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter(); gm.esdf_update(ecfg, batch=True, clear_updated_flag=True); tb = (time.perf_counter() - t0) * 1e3
    print("cap", os.environ.get("VBX_RP_MAX_ITERS", "default"), "rep", rep, "first", round(ts[0], 1), "next", [round(t, 1) for t in ts[1:]], "batch", round(tb, 1), flush=True)
PY
for cap in ${2:-128 32 40 48 56 64 128}; do
  VBX_RP_MAX_ITERS=$cap timeout 200 python /tmp/cap_probe.py 2>&1 | grep '^cap' | tee -a $OUT/cap.log
done
