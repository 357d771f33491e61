#!/bin/bash
# Winograd decoder form (csrc/srt_nn4.hip): parity at oracle sizes (forced for small batches) and per-layer timing against the direct kernels.
set -u
OUT=gpurun_out/wino; mkdir -p $OUT; export TMPDIR=/tmp
export SPLEETERRT_LIB=$PWD/spleeterrt_amd/libspleeterrt_amd_tuning.so
SRT_TUNE=wino=${1:-31},winoforce=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "test_forward_layers and mfma or test_forward_geometry_sweep or test_forward_other_geometries" 2>&1 | tail -15 | tee $OUT/parity.txt
unset SPLEETERRT_LIB
bash scripts/gpu_tune.sh wino f32 "${2:-SRT_TUNE=wino=31}" 2>&1 | tee $OUT/tune.txt
