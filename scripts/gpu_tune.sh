#!/bin/bash
# One gpurun call: quick parity subset + headline bench + tuning variants.  bash scripts/gpu_tune.sh <tag> "<SRT_TUNE settings separated by ;>"
set -u
TAG=${1:-r02b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "forward_layers or separate_end_to_end or fp16_mfma or full_size or geometry" ) > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json")); print("default", round(d["ms_per_step"],3), d["kernel_ms"])
PY
export SPLEETERRT_LIB=$PWD/spleeterrt_amd/libspleeterrt_amd_tuning.so
IFS=';' read -ra SETS <<< "${2:-}"
for t in "${SETS[@]}"; do
  SRT_TUNE="$t" timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $OUT/bench_$t.json 2>> $OUT/bench.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$t.json")); print("$t".ljust(20), round(d["ms_per_step"],3), d["kernel_ms"])
except Exception as e: print("$t", "failed", e)
PY
done
