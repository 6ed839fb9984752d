#!/bin/bash
# Round 4, on the GPU box from the repo root: rocprofv3 kernel stats of the driver-shaped command and of the reference-order
# ESDF stream.  Writes under gpurun_out/profiles_r04/ (the summaries are copied into profiles/ afterwards).
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/profiles_r04
rm -rf $OUT; mkdir -p $OUT
cd /tmp
COMMON="--no-cpu-baseline --no-extras --no-host-path --mirror-frames 0 --profile-frames 0"
rm -rf /tmp/p_fast /tmp/p_esdf
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_fast -- python $R/bench.py $COMMON --steps 20 --warmup 5 > $OUT/fast_bench.log 2>&1
cp /tmp/p_fast/*/*kernel_stats.csv $OUT/fast_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_esdf -- python $R/tools/time_esdf_strict.py 10 > $OUT/esdf_ref_order.log 2>&1
cp /tmp/p_esdf/*/*kernel_stats.csv $OUT/esdf_ref_order_kernel_stats.csv
cd $R
VBX_RP_STATS=1 python tools/time_esdf_strict.py 10 > $OUT/esdf_ref_order_phases.txt 2>&1
ls -la $OUT
