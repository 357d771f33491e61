#!/bin/bash
# round 4, call f: the peeled first K step (C = 0) in the 32-channel Winograd kernels, A/B on one box
set -u
OUT=gpurun_out/r04f; mkdir -p $OUT; export TMPDIR=/tmp
bash scripts/gpu_tune.sh r04f f32 "SRT_TUNE=winopeel=1;SRT_TUNE=winopeel=1"
