#!/bin/bash
# round 4, call m: opaque LDS operand bases in the direct encoder kernels: parity of every path that uses them + timing
set -u
OUT=gpurun_out/r04m; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "forward_layers or other_geometries or geometry_sweep or split_k or odd_geometries or down1_streamed or batch_invariant or fp16_container" ) > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
bash scripts/gpu_r04.sh r04m benchq
bash scripts/gpu_r04.sh r04m benchq
