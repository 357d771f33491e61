#!/bin/bash
# round 4, call j: fp16-storage streamed up6 - parity + the f16 bench
set -u
OUT=gpurun_out/r04j; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "up6_streamed or fp16_mfma or config4" ) > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --precision f16 > $OUT/bench_f16.json 2>> $OUT/bench.err
python - <<PY
import json
d = json.load(open("$OUT/bench_f16.json")); print("f16 ms/step", round(d["ms_per_step"], 3), d["kernel_ms"]); print(d["roofline"]["bound"], d["roofline"]["frac"], d["layer_kernels"]["up6"])
PY
tail -3 $OUT/bench.err
