#!/bin/bash
# round 6: the GPU suite + the driver's bench command at one commit
set -x
OUT=gpurun_out/${1:-r06b}; mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.log 2> $OUT/bench.err
cp bench_detail.json $OUT/ 2>/dev/null
tail -1 $OUT/bench.log | cut -c1-3000
