#!/bin/bash
OUT=gpurun_out/${1:-r06j}; mkdir -p $OUT
run() { echo "== $*"; env "$@" python tools/time_esdf_strict.py 10 2>&1 | grep "frame 0\|median" ; }
{
run VBX_RP_GRID=1024
run VBX_RP_GRID=768
run VBX_RP_GRID=512
run VBX_RP_GRID=256
run VBX_RP_KMAX=8192
run VBX_RP_KMAX=32768
run VBX_RP_SMAX=512
run VBX_RP_EV=128
run VBX_RP_MAX_ITERS=32
run VBX_RP_MAX_ITERS=128
} > $OUT/sweep.log 2>&1
cat $OUT/sweep.log
