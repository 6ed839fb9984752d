#!/bin/bash
OUT=gpurun_out/r06u; mkdir -p $OUT
python -m pytest tests/test_gpu_tsdf_parity.py tests/test_golden_reference_digests.py tests/test_gpu_block_order.py -x -q 2>&1 | tail -2
python tools/kernel_table.py 0.05 20 5 fast 2>&1 | grep "k_fold_direct\|^sum\|total_ms"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | cut -c1-200
