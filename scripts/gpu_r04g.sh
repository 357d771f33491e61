#!/bin/bash
# round 4, call g: aligned b128 patch reads in the Winograd encoder + fp16-storage streamed down1: parity, timing, f16 bench
set -u
OUT=gpurun_out/r04g; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "down1_streamed or odd_geometries or shipped_launch or fp16_mfma or config4" ) > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
bash scripts/gpu_r04.sh r04g benchq
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --precision f16 > $OUT/bench_f16.json 2>> $OUT/bench.err
python - <<PY
import json
d = json.load(open("$OUT/bench_f16.json")); print("f16 ms/step", round(d["ms_per_step"], 3), d["kernel_ms"]); print(d["roofline"]["bound"], d["roofline"]["frac"], d["layer_kernels"]["down1"])
PY
