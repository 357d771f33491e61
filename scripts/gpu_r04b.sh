#!/bin/bash
# round 4, call b: encoder epilogue v2 - parity (quick tests), timing vs the ablations, up6 / head per-tile time against the batch size (Infinity Cache)
set -u
OUT=gpurun_out/r04b; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_multi_device.py tests/test_gpu_parity.py -m gpu -q -x -k "several_gpu or multi_engine or odd_geometries or shipped_launch" ) > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
bash scripts/gpu_tune.sh r04b f32 "SRT_TUNE=encabl=6;SRT_TUNE=encabl=1;SRT_TUNE=encabl=4;SRT_TUNE=encabl=3"
for t in 8 16 32; do
  timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --tiles $t > $OUT/tiles_$t.json 2>> $OUT/bench.err
  python - <<PY
import json
d = json.load(open("$OUT/tiles_$t.json")); k = d["kernel_ms"]; print("tiles $t ms/step", round(d["ms_per_step"], 3), "per 64 tiles:", {n: round(v * 64 / $t, 3) for n, v in k.items()})
PY
done
