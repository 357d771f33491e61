#!/bin/bash
set -u
OUT=gpurun_out/r03d; mkdir -p $OUT; export TMPDIR=/tmp
export SPLEETERRT_LIB=$PWD/spleeterrt_amd/libspleeterrt_amd_tuning.so
for t in "wino32=1" "wino32=1,winoring=52,winosb=1" "winosb=1"; do
  ( SRT_TUNE=$t timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "winograd" ) > $OUT/parity_$t.log 2>&1; echo "parity $t: $(tail -1 $OUT/parity_$t.log)"
done
unset SPLEETERRT_LIB
bash scripts/gpu_tune.sh r03d f32 "SRT_TUNE=wino32=1;SRT_TUNE=wino32=1,winosb=1;SRT_TUNE=wino32=1,winoring=52;SRT_TUNE=wino32=1,winoring=52,winosb=1;SRT_TUNE=wino32=1,winoring=42;SRT_TUNE=winosb=1;SRT_TUNE=wino32=1,winoabl=1;SRT_TUNE=wino32=1,winoabl=3;SRT_TUNE=wino32=1,winoabl=4"
