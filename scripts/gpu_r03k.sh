#!/bin/bash
# Per-layer time per tile as a function of the batch (is a layer bound by HBM?  At 8 tiles x 4 stems the activations of a layer fit the 256 MiB
# Infinity Cache, at 64 x 4 they do not).
set -u
TAG=${1:-r03k}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for t in 8 16 32 64 128; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --tiles $t > $OUT/bench_t$t.json 2>> $OUT/bench.err
  python - <<PY
import json
d = json.load(open("$OUT/bench_t$t.json")); k = d["kernel_ms"]
print("tiles $t".ljust(10), round(d["ms_per_step"] / $t * 64, 3), {n: round(v / $t * 64, 3) for n, v in k.items()})
PY
done
