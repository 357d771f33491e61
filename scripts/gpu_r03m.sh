#!/bin/bash
# streamed up6: parity (new test + the forward tests with the form forced on in the tuning library), timing against the old kernel
set -u
TAG=${1:-r03m}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -s -k "up6_streamed" 2>&1 | tail -8 | tee $OUT/parity.log
SPLEETERRT_LIB=$PWD/spleeterrt_amd/libspleeterrt_amd_tuning.so SRT_TUNE_UP6=11 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "forward_layers or geometry_sweep or other_geometries" 2>&1 | tail -4 | tee -a $OUT/parity.log
bash scripts/gpu_tune.sh $TAG f32 "SRT_TUNE_UP6=10"
