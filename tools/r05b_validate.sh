#!/bin/bash
# Round 5, last GPU call (through gpurun, from the repo root): the full GPU suite at HEAD, the reference-order ESDF timed with
# and without the two changes of this session (VBX_RP_MARK_MOVED, VBX_RP_FOLD_ALL, VBX_RP_TGT_CLAIM), its phase counters, and the driver's
# command.  Everything lands in gpurun_out/r05b/.
export TMPDIR=/tmp
O=gpurun_out/${1:-r05b}
mkdir -p $O
timeout 700 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log
for mm in 2 1; do
  VBX_RP_MARK_MOVED=$mm timeout 120 python tools/time_esdf_strict.py 14 > $O/esdf_time_mark$mm.log 2>&1
done
# the same library built with -DRP_EVQ=8 (tools/_evq8/, built by hand: hipcc ... -DRP_EVQ=8 voxblox_amd/csrc/vbx_hip.hip): 512 events per target
if [ -f tools/_evq8/libvbx_hip.so ]; then
  VBX_HIP_LIB=$PWD/tools/_evq8/libvbx_hip.so VBX_RP_EV=512 timeout 120 python tools/time_esdf_strict.py 14 > $O/esdf_time_ev512.log 2>&1
  VBX_HIP_LIB=$PWD/tools/_evq8/libvbx_hip.so VBX_RP_EV=512 timeout 200 python -m pytest tests/test_gpu_esdf_reference_order.py -x -q > $O/gpu_tests_ev512.log 2>&1; echo "rc=$?" >> $O/gpu_tests_ev512.log
fi
VBX_RP_STATS=1 timeout 120 python tools/time_esdf_strict.py 12 > $O/esdf_ref_order_phases.txt 2>&1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-threads 1,16 --detail-out $O/bench_detail.json > $O/bench_line.json 2> $O/bench_err.log
echo "bench rc=$?" >> $O/gpu_tests.log
tail -3 $O/gpu_tests.log; tail -3 $O/gpu_tests_ev512.log; for f in $O/esdf_time_*.log; do echo $f; grep 'frame 0 ' $f; tail -1 $f; done; wc -c $O/bench_line.json
