#!/bin/bash
# Round 6, on the GPU box from the repo root (through gpurun), at one commit: rocprofv3 kernel stats of the driver-shaped command,
# of the configs[3] leg in reference order and of configs[4] on whole-sensor bundles; FETCH_SIZE / WRITE_SIZE passes (separate runs,
# --kernel-trace only) of the driver-shaped command and of the reference-order ESDF stream; the phase counters of the first
# updates of the stream; the driver's bench line.  Raw CSVs -> gpurun_out/profiles_new/, summaries -> profiles/r06_* (copied to
# gpurun_out/profiles_r06_summaries/ so that they come back).
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
python tools/summarize_profiles.py $OUT r06 > $OUT/summary.txt 2>&1
python tools/summarize_esdf_pmc.py r06 >> $OUT/summary.txt 2>&1
VBX_RP_STATS=1 python tools/time_esdf_strict.py 6 > profiles/r06_esdf_ref_order_phases.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.log 2> $OUT/bench_line.err
tail -1 $OUT/bench_line.log > profiles/r06_bench_line.json
cp bench_detail.json profiles/r06_bench_detail.json
mkdir -p gpurun_out/profiles_r06_summaries; cp profiles/r06_* gpurun_out/profiles_r06_summaries/ 2>/dev/null
rm -f $OUT/pmc_*_counter_collection.csv   # (tens of MB; the summaries are what is kept)
ls -la gpurun_out/profiles_r06_summaries; tail -1 $OUT/bench_line.log | cut -c1-400
