#!/bin/bash
# Winograd decoder form (csrc/srt_nn4.hip), one gpurun call:  bash scripts/gpu_wino.sh <layer mask> "SRT_TUNE=wino=30;SRT_TUNE=wino=30,winoabl=4;..."
# 1. parity at oracle sizes with the form FORCED on for small batches too (tuning library, SRT_TUNE=wino=<mask>,winoforce=1): every tensor of
#    test_forward_layers / geometry sweeps against the CPU oracle;  2. per-layer timing of the listed settings against the product library.
set -u
OUT=gpurun_out/wino; mkdir -p $OUT; export TMPDIR=/tmp
export SPLEETERRT_LIB=$PWD/spleeterrt_amd/libspleeterrt_amd_tuning.so
SRT_TUNE=wino=${1:-31},winoforce=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "test_forward_layers and mfma or test_forward_geometry_sweep or test_forward_other_geometries" 2>&1 | tail -15 | tee $OUT/parity.txt
unset SPLEETERRT_LIB
bash scripts/gpu_tune.sh wino f32 "${2:-SRT_TUNE=wino=31}" 2>&1 | tee $OUT/tune.txt
