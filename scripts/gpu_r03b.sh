#!/bin/bash
# wino32: parity (tuning library, SRT_TUNE=wino32=1) + timings + ablations; microbench rows
set -u
OUT=gpurun_out/r03b; mkdir -p $OUT; export TMPDIR=/tmp
timeout 200 scripts/ubench/mfma_valu > $OUT/ubench.log 2>&1; tail -22 $OUT/ubench.log
export SPLEETERRT_LIB=$PWD/spleeterrt_amd/libspleeterrt_amd_tuning.so
( SRT_TUNE=wino32=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "winograd or batch_invariant" ) > $OUT/parity.log 2>&1; grep -v "^$" $OUT/parity.log | tail -25
unset SPLEETERRT_LIB
bash scripts/gpu_tune.sh r03b f32 "SRT_TUNE=wino32=1;SRT_TUNE=wino32=1,winoabl=1;SRT_TUNE=wino32=1,winoabl=3;SRT_TUNE=wino32=1,winoabl=4;SRT_TUNE=wino32=1,winoabl=5;SRT_TUNE=wino32=1,winotpw=1;SRT_TUNE=wino32=1,winotpw=2"
