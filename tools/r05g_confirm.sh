#!/bin/bash
# Round 5, last GPU seconds: the reference-order tests and one timing with the defaults as shipped (VBX_RP_LDS_COUNTS = 1).
export TMPDIR=/tmp
O=gpurun_out/r05g
mkdir -p $O
timeout 60 python -m pytest tests/test_gpu_esdf_reference_order.py -x -q > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log
timeout 30 python tools/time_esdf_strict.py 14 > $O/esdf_time.log 2>&1
tail -3 $O/gpu_tests.log; grep 'frame 0 ' $O/esdf_time.log; tail -1 $O/esdf_time.log
