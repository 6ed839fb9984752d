export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/profiles_new2
rm -rf $OUT; mkdir -p $OUT
cd /tmp
COMMON="--no-cpu-baseline --no-extras --no-host-path --mirror-frames 0 --profile-frames 0"
rm -rf /tmp/p_s4
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_s4 -- python $R/bench.py $COMMON --workload sensors4 --steps 3 --warmup 2 > $OUT/sensors4_bench.log 2>&1
cp /tmp/p_s4/*/*kernel_stats.csv $OUT/sensors4_kernel_stats.csv
cd $R
python bench.py --workload sensors4 --gpus 1 --steps 6 --warmup 2 > $OUT/bench_sensors4_1gpu.json 2> $OUT/bench_sensors4_1gpu.err
python bench.py --workload sensors4 --gpus 1 --steps 6 --warmup 2 --exchange torch --no-cpu-baseline > $OUT/bench_sensors4_1gpu_torch.json 2> $OUT/bench_sensors4_1gpu_torch.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_shaped.json 2> $OUT/bench_driver_shaped.err
ls -la $OUT
