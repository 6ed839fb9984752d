#!/bin/bash
# Round 5, last GPU seconds: VBX_RP_LDS_COUNTS=1 (per-workgroup counts in the commit folds) through every test that runs the
# reference-order ESDF, then timed against 0.
export TMPDIR=/tmp
O=gpurun_out/r05f
mkdir -p $O
VBX_RP_LDS_COUNTS=1 timeout 150 python -m pytest tests/test_gpu_esdf_reference_order.py tests/test_gpu_block_order.py tests/test_gpu_esdf_parity.py tests/test_gpu_dropin_real_headers.py tests/test_gpu_dropin_host_edits.py -x -q > $O/gpu_tests_lds1.log 2>&1; echo "rc=$?" >> $O/gpu_tests_lds1.log
for v in 1 0; do
  VBX_RP_LDS_COUNTS=$v timeout 60 python tools/time_esdf_strict.py 14 > $O/esdf_time_lds$v.log 2>&1
done
VBX_RP_LDS_COUNTS=1 VBX_RP_STATS=1 timeout 60 python tools/time_esdf_strict.py 6 > $O/esdf_phases_lds1.txt 2>&1
tail -3 $O/gpu_tests_lds1.log; for f in $O/esdf_time_*.log; do echo $f; grep 'frame 0 ' $f; tail -1 $f; done
