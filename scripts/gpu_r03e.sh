#!/bin/bash
set -u
OUT=gpurun_out/r03e; mkdir -p $OUT; export TMPDIR=/tmp
export SPLEETERRT_LIB=$PWD/spleeterrt_amd/libspleeterrt_amd_tuning.so
for t in "wino32=1,winosb=1,winoea=1" "wino32=1,winosb=1,winoea=2"; do
  ( SRT_TUNE=$t timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "winograd" ) > $OUT/parity_$t.log 2>&1; echo "parity $t: $(tail -1 $OUT/parity_$t.log)"
done
( SRT_TUNE_UP6=6 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "forward_layers or geometry_sweep" ) > $OUT/parity_up6.log 2>&1; echo "parity up6=6: $(tail -1 $OUT/parity_up6.log)"
unset SPLEETERRT_LIB
bash scripts/gpu_tune.sh r03e f32 "SRT_TUNE=wino32=1,winosb=1;SRT_TUNE=wino32=1,winosb=1,winoea=1;SRT_TUNE=wino32=1,winosb=1,winoea=2;SRT_TUNE_UP6=6;SRT_TUNE_UP6=7;SRT_TUNE_UP6=8;SRT_TUNE_UP6=9"
