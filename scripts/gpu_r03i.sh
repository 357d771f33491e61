#!/bin/bash
# wino16 (class-split 16-channel Winograd kernel for up5): parity under the tuning library, then timing against the product.
set -u
TAG=${1:-r03i}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
SPLEETERRT_LIB=$PWD/spleeterrt_amd/libspleeterrt_amd_tuning.so SRT_TUNE=wino16=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "winograd or layer_taps or full_forward or odd" 2>&1 | tail -8 | tee $OUT/parity.log
bash scripts/gpu_tune.sh $TAG f32 "SRT_TUNE=wino16=1"
