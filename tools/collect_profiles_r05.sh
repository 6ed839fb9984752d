#!/bin/bash
# Round 5, on the GPU box from the repo root (through gpurun).  rocprofv3 kernel stats of the driver-shaped command, of the
# configs[3] leg in reference order and of configs[4] on whole-sensor bundles; FETCH_SIZE / WRITE_SIZE passes (separate runs,
# --kernel-trace only) of the driver-shaped command and of the reference-order ESDF stream.  Raw CSVs -> gpurun_out/profiles_new/
# (+ pmc_esdf_*), summaries -> profiles/r05_* by tools/summarize_profiles.py and the python block at the end.
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/profiles_new
rm -rf $OUT; mkdir -p $OUT
cd /tmp
COMMON="--no-cpu-baseline --no-extras --no-host-path --mirror-frames 0 --profile-frames 0"
stats() {  # name, bench args...
  local name=$1; shift
  rm -rf /tmp/p_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -- python $R/bench.py $COMMON "$@" --detail-out $OUT/${name}_detail.json > $OUT/${name}_bench.log 2>&1
  cp /tmp/p_$name/*/*kernel_stats.csv $OUT/${name}_kernel_stats.csv
}
stats fast --steps 20 --warmup 5
stats esdf --esdf --steps 20 --warmup 3
stats sensors4 --workload sensors4 --steps 2 --warmup 1
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$C /tmp/pe_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/p_$C -- python $R/bench.py $COMMON --steps 20 --warmup 5 --detail-out /tmp/d.json > $OUT/pmc_$C.log 2>&1
  cp /tmp/p_$C/*/*counter_collection.csv $OUT/pmc_${C}_counter_collection.csv
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pe_$C -- python $R/bench.py $COMMON --esdf --steps 10 --warmup 3 --detail-out /tmp/d.json > $OUT/pmc_esdf_$C.log 2>&1
  cp /tmp/pe_$C/*/*counter_collection.csv $OUT/pmc_esdf_${C}_counter_collection.csv
done
cd $R
python tools/summarize_profiles.py $OUT r05 > $OUT/summary.txt 2>&1
python tools/summarize_esdf_pmc.py

mkdir -p gpurun_out/profiles_r05_summaries; cp profiles/r05_* gpurun_out/profiles_r05_summaries/ 2>/dev/null
rm -f $OUT/pmc_*_counter_collection.csv   # (tens of MB; the summaries are what is kept)
ls -la $OUT gpurun_out/profiles_r05_summaries
