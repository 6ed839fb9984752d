#!/bin/bash
# Run on the GPU box from the repo root: rocprofv3 kernel stats for every bench workload, the two PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate runs, --kernel-trace only) for the default workload, and the bench lines.
# Writes under gpurun_out/profiles_new/ (copied into profiles/ afterwards by tools/summarize_profiles.py).
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/profiles_new
rm -rf $OUT; mkdir -p $OUT
cd /tmp
COMMON="--no-cpu-baseline --no-extras --no-host-path --mirror-frames 0 --profile-frames 0"
stats() {  # name, bench args...
  local name=$1; shift
  rm -rf /tmp/p_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -- python $R/bench.py $COMMON "$@" > $OUT/${name}_bench.log 2>&1
  cp /tmp/p_$name/*/*kernel_stats.csv $OUT/${name}_kernel_stats.csv
}
stats fast --steps 20 --warmup 5
stats esdf --esdf --steps 20 --warmup 3 --esdf-fidelity-frames 0
stats merged_cow --integrator merged --scene cow --steps 20 --warmup 3
stats simple --integrator simple --steps 6 --warmup 2
stats sensors4 --workload sensors4 --steps 2 --warmup 1
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$C
  # the driver-shaped command (20 timed steps after 5 warm-up steps): the summary keeps the timed frames only
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/p_$C -- python $R/bench.py $COMMON --steps 20 --warmup 5 > $OUT/pmc_$C.log 2>&1
  cp /tmp/p_$C/*/*counter_collection.csv $OUT/pmc_${C}_counter_collection.csv
done
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --esdf --no-extras > $OUT/bench_esdf.json 2> $OUT/bench_esdf.err
python bench.py --integrator merged --scene cow --no-extras > $OUT/bench_merged_cow.json 2> $OUT/bench_merged_cow.err
python bench.py --integrator simple --no-extras --steps 10 --warmup 2 > $OUT/bench_simple.json 2> $OUT/bench_simple.err
python bench.py --workload sensors4 --gpus 1 --steps 6 --warmup 2 > $OUT/bench_sensors4_1gpu.json 2> $OUT/bench_sensors4_1gpu.err
python bench.py --mesh --no-extras > $OUT/bench_mesh.json 2> $OUT/bench_mesh.err
# kernel timelines (one frame / one ESDF update, with the idle gap before every kernel)
bash tools/frame_timeline.sh 0.05 24 20 > $OUT/timeline_0p05.txt 2>&1
bash tools/frame_timeline.sh 0.02 6 4 > $OUT/timeline_0p02.txt 2>&1
bash tools/esdf_timeline.sh 24 20 > $OUT/timeline_esdf.txt 2>&1
ls -la $OUT
