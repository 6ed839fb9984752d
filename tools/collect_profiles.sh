#!/bin/bash
# Run on the GPU box from the repo root: kernel stats + the two PMC passes + a bench line.
# Writes everything under gpurun_out/profiles_new/ (copied into profiles/ afterwards).
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/profiles_new
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rm -rf /tmp/p_stats /tmp/p_fetch /tmp/p_write
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --mirror-frames 0 --no-variants > $OUT/stats_bench.log 2>&1
cp /tmp/p_stats/*/*kernel_stats.csv $OUT/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_fetch -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --mirror-frames 0 --no-variants > $OUT/fetch_bench.log 2>&1
cp /tmp/p_fetch/*/*counter_collection.csv $OUT/pmc_fetch_size_counter_collection.csv
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_write -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --mirror-frames 0 --no-variants > $OUT/write_bench.log 2>&1
cp /tmp/p_write/*/*counter_collection.csv $OUT/pmc_write_size_counter_collection.csv
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
ls -la $OUT
