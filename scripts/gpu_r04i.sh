#!/bin/bash
# round 4, call i: patch ring of four in the Winograd encoder (patches three K steps ahead): parity under the tuning library, timing A/B
set -u
OUT=gpurun_out/r04i; mkdir -p $OUT; export TMPDIR=/tmp
( SPLEETERRT_LIB=$PWD/spleeterrt_amd/libspleeterrt_amd_tuning.so SRT_TUNE=winopr=4 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "odd_geometries" ) > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
bash scripts/gpu_tune.sh r04i f32 "SRT_TUNE=winopr=4;SRT_TUNE=winopr=4"
