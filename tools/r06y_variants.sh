#!/bin/bash
# round 6: the reference-order tests under the measurement switches that change the step kernel's paths (grid not a multiple of 32 and
# smaller than the tile count of a big scan, one target-id counter, the serial debug form)
OUT=gpurun_out/${1:-r06y}; mkdir -p $OUT
export TMPDIR=/tmp
for V in "VBX_RP_GRID=96" "VBX_RP_GRID=200 VBX_RP_TGT_SHARDS=1" "VBX_RP_TGT_SHARDS=3 VBX_RP_KMAX=4096" "VBX_RP_SMAX=128"; do
  echo "== $V"
  ( env $V timeout 600 python -m pytest tests/test_gpu_esdf_reference_order.py -x -q 2>&1 | tail -2 )
done 2>&1 | tee $OUT/variants.log
