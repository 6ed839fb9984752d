#!/bin/bash
# Round 5, after the last changes to the reference-order ESDF (rankings mark moved records, no dirty marks in PLACE_BASE, targets
# claimed before they take an id, 256 events per target): rocprofv3 kernel stats and FETCH_SIZE / WRITE_SIZE passes of the
# configs[3] leg only (the TSDF kernels did not change: profiles/r05_fast_* stand).  Same commands as tools/collect_profiles_r05.sh.
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/profiles_new
rm -rf $OUT; mkdir -p $OUT
cd /tmp
COMMON="--no-cpu-baseline --no-extras --no-host-path --mirror-frames 0 --profile-frames 0"
rm -rf /tmp/p_esdf
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_esdf -- python $R/bench.py $COMMON --esdf --steps 20 --warmup 3 --detail-out $OUT/esdf_detail.json > $OUT/esdf_bench.log 2>&1
cp /tmp/p_esdf/*/*kernel_stats.csv $OUT/esdf_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pe_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pe_$C -- python $R/bench.py $COMMON --esdf --steps 10 --warmup 3 --detail-out /tmp/d.json > $OUT/pmc_esdf_$C.log 2>&1
  cp /tmp/pe_$C/*/*counter_collection.csv $OUT/pmc_esdf_${C}_counter_collection.csv
done
cd $R
mkdir -p gpurun_out/profiles_out
python tools/summarize_profiles.py $OUT r05 > $OUT/summary.txt 2>&1
python tools/summarize_esdf_pmc.py >> $OUT/summary.txt 2>&1
cp profiles/r05_esdf_kernel_stats.md profiles/r05_esdf_kernel_stats.csv profiles/r05_pmc_esdf_ref_order.json gpurun_out/profiles_out/
rm -f $OUT/pmc_*_counter_collection.csv   # (tens of MB; the summaries are what is kept)
tail -5 $OUT/esdf_bench.log | cut -c1-400; head -14 profiles/r05_esdf_kernel_stats.md
