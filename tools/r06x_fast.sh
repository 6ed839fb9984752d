#!/bin/bash
# round 6: the Fast path after a change — its parity tests, then the driver-shaped bench line without the extras
OUT=gpurun_out/${1:-r06x}; mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_tsdf_parity.py tests/test_gpu_block_order.py tests/test_gpu_sensors4_parity.py tests/test_gpu_cpp_shim.py -x -q ) > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-host-path --mirror-frames 0 > $OUT/bench.log 2> $OUT/bench.err
tail -1 $OUT/bench.log | cut -c1-700
