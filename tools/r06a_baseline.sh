#!/bin/bash
# round 6, first GPU call: the suite at HEAD on today's box + the diagnostics the round's perf work starts from
set -x
OUT=gpurun_out/r06a; mkdir -p $OUT
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
VBX_DEBUG=1 python tools/kernel_table.py 0.05 25 0 fast > $OUT/fast_debug.log 2>&1
VBX_RP_STATS=1 python tools/time_esdf_strict.py 8 > $OUT/esdf_phases.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.log 2>&1
cp bench_detail.json $OUT/ 2>/dev/null
tail -3 $OUT/pytest.log; tail -1 $OUT/bench.log | cut -c1-600
