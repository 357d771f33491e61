#!/bin/bash
# A/B of library variants on ONE box:  bash scripts/gpu_libs_ab.sh <tag> "<bench args>" <lib.so | default> ...
# (each variant = the shipped objects with one source file rebuilt under other -D flags, e.g. scratch/pd/libsrt_pd_3_3_1.so; `default` = the in-tree library)
set -u
TAG=$1; ARGS=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
for lib in "$@"; do
  lab=$(basename $lib .so)
  if [ "$lib" = default ]; then envs="SRT_AB=0"; else envs="SPLEETERRT_LIB=$lib"; fi
  for rep in 1 2; do
    env $envs timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $ARGS > $OUT/bench_${lab}_$rep.json 2>> $OUT/bench.err
    python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_${lab}_$rep.json")); print("$lab".ljust(22), round(d["ms_per_step"], 3), round(sum(d["kernel_ms"].values()), 3), d["kernel_ms"])
except Exception as e:
    print("$lab", "failed", e)
PY
  done
done
