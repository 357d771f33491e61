#!/bin/bash
# round 4, call e: first K step with C = 0 (no accumulator zeroing) - parity of the Winograd kernels + timing
set -u
OUT=gpurun_out/r04e; mkdir -p $OUT; export TMPDIR=/tmp
bash scripts/gpu_r04.sh r04e quick benchq
