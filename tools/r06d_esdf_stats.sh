#!/bin/bash
set -x
OUT=gpurun_out/${1:-r06d}; mkdir -p $OUT
VBX_RP_STATS=1 python tools/time_esdf_strict.py ${2:-5} > $OUT/esdf_phases.log 2>&1
grep -v "^\[cls\]" $OUT/esdf_phases.log | cut -c1-700
