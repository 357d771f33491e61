#!/bin/bash
set -u
TAG=${1:-r02i}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( time SRT_DUAL=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "full_size or forward_layers or geometry_sweep" ) > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
for d in 0 1 1; do
SRT_DUAL=$d timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_dual$d.json 2>> $OUT/bench.err
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_dual$d.json")); print("dual=$d", round(d["ms_per_step"],3), d["kernel_ms"])
except Exception as e: print("dual=$d failed", e)
PY
done
