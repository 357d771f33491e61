#!/bin/bash
set -u
TAG=${1:-r02g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fp16 or config4 or forward_layers" ) > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
for prec in f16; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --precision $prec > $OUT/bench_$prec.json 2>> $OUT/bench.err
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$prec.json")); print("$prec", round(d["ms_per_step"],3), d["kernel_ms"])
except Exception as e: print("$prec failed", e)
PY
done
export SPLEETERRT_LIB=$PWD/spleeterrt_amd/libspleeterrt_amd_tuning.so
IFS=';' read -ra SETS <<< "${2:-}"
for t in "${SETS[@]}"; do
  SRT_TUNE="$t" timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > "$OUT/bench_$t.json" 2>> $OUT/bench.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$t.json")); print("$t".ljust(20), round(d["ms_per_step"],3), d["kernel_ms"])
except Exception as e: print("$t", "failed", e)
PY
done
