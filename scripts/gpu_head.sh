#!/bin/bash
set -u
OUT=gpurun_out/${1:-r02o}; mkdir -p $OUT; export TMPDIR=/tmp
export SPLEETERRT_LIB=$PWD/spleeterrt_amd/libspleeterrt_amd_tuning.so
for v in 1 2 4 8; do
  SRT_TUNE_HEAD=$v timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_head$v.json 2>> $OUT/bench.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_head$v.json")); print("head=$v", round(d["ms_per_step"],3), d["kernel_ms"]["up7"])
PY
done
