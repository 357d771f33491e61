#!/bin/bash
# round 4, call c: streamed down1 (parity + timing vs the tiled kernel + ablations), what the Winograd encoder's unit epilogue costs (ablations)
set -u
OUT=gpurun_out/r04c; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "down1_streamed or shipped_launch" ) > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
bash scripts/gpu_tune.sh r04c f32 "SRT_TUNE=d1s=0;SRT_TUNE=d1s=2;SRT_TUNE=d1s=3;SRT_TUNE=encabl=6;SRT_TUNE=encabl=7;SRT_TUNE=encabl=8;SRT_TUNE=encabl=9"
