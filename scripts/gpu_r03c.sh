#!/bin/bash
# ring depth / priority experiments for both Winograd kernels + parity of the variants
set -u
OUT=gpurun_out/r03c; mkdir -p $OUT; export TMPDIR=/tmp
export SPLEETERRT_LIB=$PWD/spleeterrt_amd/libspleeterrt_amd_tuning.so
for t in "wino32=1" "wino32=1,winoring=5" "winoring=5"; do
  ( SRT_TUNE=$t timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "winograd" ) > $OUT/parity_$t.log 2>&1; echo "parity $t: $(tail -1 $OUT/parity_$t.log)"
done
unset SPLEETERRT_LIB
bash scripts/gpu_tune.sh r03c f32 "SRT_TUNE=wino32=1;SRT_TUNE=wino32=1,winoring=4;SRT_TUNE=wino32=1,winoring=5;SRT_TUNE=winoring=4;SRT_TUNE=winoring=5;SRT_TUNE=winoring=6;SRT_TUNE=wino32=1,winoprio=1;SRT_TUNE=wino32=1,winoring=5,winoprio=1"
