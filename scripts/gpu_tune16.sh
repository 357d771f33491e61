#!/bin/bash
set -u
TAG=${1:-r02n}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export SPLEETERRT_LIB=$PWD/spleeterrt_amd/libspleeterrt_amd_tuning.so
for v in 0 1 2; do
  SRT_TUNE16=$v timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --precision f16 > $OUT/bench_f16_v$v.json 2>> $OUT/bench.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_f16_v$v.json")); print("tune16=$v", round(d["ms_per_step"],3), d["kernel_ms"])
except Exception as e: print("tune16=$v failed", e)
PY
done
