#!/bin/bash
# round 4, call d: launch order (M-block pair fastest), patch locality ablation, barrier-per-two-steps arrangement, units per workgroup
set -u
OUT=gpurun_out/r04d; mkdir -p $OUT; export TMPDIR=/tmp
bash scripts/gpu_tune.sh r04d f32 "SRT_TUNE=winomf=1;SRT_TUNE=encabl=10;SRT_TUNE=winocfg=8;SRT_TUNE=winotpw=16;SRT_TUNE=winomf=1,winocfg=8"
