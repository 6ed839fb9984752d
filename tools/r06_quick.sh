#!/bin/bash
# round 6, last session: ESDF reference-order tests + the stream's update times (24 frames) at one build
OUT=gpurun_out/${1:-r06q}; mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_esdf_reference_order.py tests/test_gpu_esdf_parity.py -m gpu -x -q -n 4 ) > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
timeout 200 python tools/time_esdf_strict.py 24 2>&1 | grep -E "^frame" | awk '{printf "%s ", $6} END {print ""}' | tee $OUT/series.txt
timeout 200 python tools/time_esdf_strict.py 24 2>&1 | grep -E "^frame" | awk '{printf "%s ", $6} END {print ""}' | tee -a $OUT/series.txt
python - <<PY
for l in open("$OUT/series.txt"):
    v = [float(x) for x in l.split()]
    if len(v) >= 23: print("first", v[0], "mean 3..22", round(sum(v[3:23]) / 20, 2), "sum 1..11", round(sum(v[1:12]), 1))
PY
