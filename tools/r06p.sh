#!/bin/bash
OUT=gpurun_out/r06p; mkdir -p $OUT
run() { echo "== $*"; env "$@" python tools/time_esdf_strict.py 10 2>&1 | grep "frame 0\|median" ; }
{
run VBX_RP_EV=256
run VBX_HIP_LIB=$PWD/tools/_tmp/libvbx_sc12.so
run VBX_HIP_LIB=$PWD/tools/_tmp/libvbx_sc16.so
run VBX_RP_EV=256
} > $OUT/sweep.log 2>&1
cat $OUT/sweep.log
