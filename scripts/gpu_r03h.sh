#!/bin/bash
set -u
OUT=gpurun_out/r03h; mkdir -p $OUT; export TMPDIR=/tmp
export SPLEETERRT_LIB=$PWD/spleeterrt_amd/libspleeterrt_amd_tuning.so
for t in "winocfg=8,winocs=1" "winocfg=9,winocs=2"; do
  ( SRT_TUNE=$t timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "winograd" ) > $OUT/parity_$t.log 2>&1; echo "parity $t: $(tail -1 $OUT/parity_$t.log)"
done
unset SPLEETERRT_LIB
bash scripts/gpu_tune.sh r03h f32 "SRT_TUNE=winocfg=8;SRT_TUNE=winocfg=9;SRT_TUNE=winocs=1;SRT_TUNE=winocs=2;SRT_TUNE=winocfg=8,winocs=1"
