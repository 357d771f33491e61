#!/bin/bash
set -u
TAG=${1:-r02f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fp16 or config4 or forward_layers" ) > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
for prec in f16 f16x2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --precision $prec > $OUT/bench_$prec.json 2>> $OUT/bench.err
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$prec.json")); print("$prec", round(d["ms_per_step"],3), d["kernel_ms"])
except Exception as e: print("$prec failed", e)
PY
done
timeout 200 python scripts/latency_c2.py f16 > $OUT/lat_f16.log 2>&1
tail -2 $OUT/lat_f16.log | head -1
