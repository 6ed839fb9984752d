#!/bin/bash
export TMPDIR=/tmp
R=$PWD; OUT=$R/gpurun_out/r06l; mkdir -p $OUT; cd /tmp
COMMON="--no-cpu-baseline --no-extras --no-host-path --mirror-frames 0 --profile-frames 0"
rm -rf /tmp/p_esdf
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_esdf -- python $R/bench.py $COMMON --esdf --steps 10 --warmup 3 --detail-out $OUT/esdf_detail.json > $OUT/esdf_bench.log 2>&1
cp /tmp/p_esdf/*/*kernel_stats.csv $OUT/esdf_kernel_stats.csv
head -40 $OUT/esdf_kernel_stats.csv | cut -c1-160
