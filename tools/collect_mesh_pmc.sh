#!/bin/bash
# Run on the GPU box from the repo root: PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs) of
# `bench.py --mesh`, filtered to the mesher's kernels.  Output: gpurun_out/mesh_pmc/*.csv
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/mesh_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/p_$C -- python $R/bench.py --mesh --steps 5 --warmup 1 --no-cpu-baseline --mirror-frames 0 --no-variants > $OUT/$C.log 2>&1
  grep -E "Counter_Name|k_mesh" /tmp/p_$C/*/*counter_collection.csv > $OUT/pmc_$C.csv
done
ls -la $OUT
python - <<'PY'
import csv, collections, os
out = os.environ.get("R", ".") + "/gpurun_out/mesh_pmc" if False else "gpurun_out/mesh_pmc"
PY
