#!/bin/bash
set -u
OUT=gpurun_out/r03f; mkdir -p $OUT; export TMPDIR=/tmp
export SPLEETERRT_LIB=$PWD/spleeterrt_amd/libspleeterrt_amd_tuning.so
for t in "wino32=1,winosb=1,winost=1"; do
  ( SRT_TUNE=$t timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "winograd" ) > $OUT/parity_$t.log 2>&1; echo "parity $t: $(tail -1 $OUT/parity_$t.log)"
done
unset SPLEETERRT_LIB
bash scripts/gpu_tune.sh r03f f32 "SRT_TUNE=wino32=1,winosb=1;SRT_TUNE=wino32=1,winosb=1,winost=1;SRT_TUNE=wino32=1,winosb=1,winost=2;SRT_TUNE=wino32=1,winosb=1,winost=3"
