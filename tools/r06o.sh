#!/bin/bash
OUT=gpurun_out/r06o; mkdir -p $OUT
python -m pytest tests/test_gpu_dropin_real_headers.py tests/test_gpu_dropin_host_edits.py -x -q 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --cpu-threads 1 > $OUT/bench.log 2> $OUT/bench.err
python - <<'P'
import json
j=json.loads(open('gpurun_out/r06o/bench.log').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['dropin_path'])
P
