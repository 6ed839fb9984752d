#!/bin/bash
OUT=gpurun_out/${1:-r06i}; mkdir -p $OUT
python tools/time_esdf_strict.py 8 > $OUT/esdf_nostats.log 2>&1
grep "esdf ms\|median" $OUT/esdf_nostats.log
