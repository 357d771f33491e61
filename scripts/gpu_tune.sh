#!/bin/bash
# One gpurun call that measures kernel variants of the tuning library (python -m spleeterrt_amd.build --tuning) against the product
# library on the same box:   bash scripts/gpu_tune.sh <tag> <f32|f16|f16x2> "SRT_TUNE=decx=20;SRT_TUNE=bm128=1;SRT_TUNE16=1;SRT_TUNE_HEAD=2"
# Prints one line per setting: ms per 64-tile step and the per-layer kernel times (HIP events).  Keys: csrc/srt_nn2.hip (SRT_TUNE=key=value,...:
# down1 down2 up4 up5 abl eabl encx decx bm128 occ3 dual), csrc/srt_nn4.hip (wino=<layer mask> winoforce=1 winotpw=<units per workgroup>
# wino32=0|1 winocfg=1..10 winoabl=1..5 winoprio winoring winosb winocs winowalk=0|1 encwino=0|<min Cin> enccopy=0|1 encabl=1|3..10 winomf=0|1 winopeel=1
# winopr=4; csrc/srt_nn2.hip also d1s=0|2|3), csrc/srt_nn3.hip (SRT_TUNE16),
# csrc/srt_nn.hip (SRT_TUNE_UP6: 1-9 tile shapes of the old kernel, 10 old kernel, 11 streamed kernel forced, 12-14 its ablations; SRT_TUNE_HEAD,
# SRT_TUNE_HEADROWS=0|2|4).
set -u
export SRT_BENCH_NOCHECK=1      # ablation settings compute wrong results by construction
TAG=${1:-tune}; PREC=${2:-f32}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() {   # label, env assignment (may be empty)
  env $2 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --precision $PREC > "$OUT/bench_$1.json" 2>> $OUT/bench.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$1.json")); print("$1".ljust(28), round(d["ms_per_step"], 3), d["kernel_ms"])
except Exception as e:
    print("$1", "failed", e)
PY
}
run product ""
export SPLEETERRT_LIB=$PWD/spleeterrt_amd/libspleeterrt_amd_tuning.so
run tuning-default ""
IFS=';' read -ra SETS <<< "${3:-}"
for t in "${SETS[@]}"; do run "$t" "$t"; done
