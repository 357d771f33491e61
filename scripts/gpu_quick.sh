#!/bin/bash
set -u
TAG=${1:-r02l}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "forward_layers or fp16 or config4 or geometry or full_size or other_geom" ) > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
for prec in f32 f16; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --precision $prec > $OUT/bench_$prec.json 2>> $OUT/bench.err
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$prec.json")); print("$prec", round(d["ms_per_step"],3), d["kernel_ms"])
except Exception as e: print("$prec failed", e)
PY
done
