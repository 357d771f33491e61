#!/bin/bash
set -u
TAG=${1:-r02c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1100 python -m pytest tests -m gpu -q --durations=8 ) > $OUT/pytest.log 2>&1
tail -12 $OUT/pytest.log
timeout 200 python scripts/latency_c2.py f32 --out $OUT/latency.json > $OUT/lat.log 2>&1
tail -3 $OUT/lat.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json")); print("default", round(d["ms_per_step"],3), d["kernel_ms"])
PY
