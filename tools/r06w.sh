#!/bin/bash
# round 6: the ranking's tick statistics only (no tests): VBX_RP_STATS over the first updates of the stream
OUT=gpurun_out/${1:-r06w}; mkdir -p $OUT
export TMPDIR=/tmp
VBX_RP_STATS=1 timeout 120 python tools/time_esdf_strict.py ${2:-5} > $OUT/esdf_phases.log 2>&1
grep -E "rankings|folds:|^frame|steps:|launches by|by duration|push tiles|control step" $OUT/esdf_phases.log | cut -c1-700
timeout 120 python tools/time_esdf_strict.py ${2:-5} 2>&1 | grep -E "^frame|median"
