#!/bin/bash
# round 6: the voxel walk's kernels after a change — ESDF parity tests, then rocprofv3 kernel stats of a short reference-order run
OUT=gpurun_out/${1:-r06z}; mkdir -p $OUT
export TMPDIR=/tmp
( timeout 500 python -m pytest tests/test_gpu_esdf_reference_order.py tests/test_gpu_esdf_parity.py -x -q ) > $OUT/pytest.log 2>&1
tail -2 $OUT/pytest.log
cd /tmp && rm -rf /tmp/p_cls && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_cls -- python $OLDPWD/tools/time_esdf_strict.py 6 > $OLDPWD/$OUT/prof.log 2>&1
cd $OLDPWD
grep -h "k_cls\|k_rp_hazard\|k_rp_nbslot" /tmp/p_cls/*/*kernel_stats.csv | cut -d, -f1-5 | cut -c1-160
grep -E "^frame" $OUT/prof.log | tail -3
