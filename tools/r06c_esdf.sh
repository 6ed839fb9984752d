#!/bin/bash
# round 6: the reference-order ESDF after a change — its 73 tests, then the phase counters and times of the stream's first updates
set -x
OUT=gpurun_out/${1:-r06c}; mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 400 python -m pytest tests/test_gpu_esdf_reference_order.py tests/test_gpu_block_order.py tests/test_gpu_esdf_parity.py tests/test_gpu_dropin_real_headers.py tests/test_gpu_dropin_host_edits.py -x -q ) > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
VBX_RP_STATS=1 timeout 120 python tools/time_esdf_strict.py ${2:-8} > $OUT/esdf_phases.log 2>&1
grep -v "^\[cls\]" $OUT/esdf_phases.log | cut -c1-600
