#!/bin/bash
# Ablation matrix of the C8-form fp16 kernels (csrc/srt_nn5.hip, SRT_TUNE_C8 bits: 1 no patch DMA, 2 no weight DMA, 4 no MFMAs, 8 no epilogue, 16 MFMAs without
# LDS reads, 32 no barrier / DMA wait) on the tuning library, 5-stem f16 bench shape:   bash scripts/gpu_abl_c8.sh <tag> "0 3 4 8 16 12"
# Only the combinations listed in C8_ABL_CASES (srt_nn5.hip) are instantiated - 3 4 8 16 12 as shipped; another value runs the unablated kernel.  (The matrix of
# DESIGN.md 3.2 also holds 1 / 2 / 32 / 19 and the epilogue-only ablations 64 / 128 / 192, measured from earlier states of the file: gpurun_out/abl_c8, abl_c8b.)
# SRT_TUNE_C8LW=1 selects the loader-wave form, SRT_TUNE_D2C8=1 times down2 on the C8 kernel.
set -u
export SRT_BENCH_NOCHECK=1
TAG=${1:-abl_c8}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
export SPLEETERRT_LIB=$PWD/spleeterrt_amd/libspleeterrt_amd_tuning.so
for a in ${2:-0 3 4 8 16 12}; do
  SRT_TUNE_C8=$a timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --precision f16 --stems 5 > $OUT/bench_$a.json 2>> $OUT/bench.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$a.json")); k = d["kernel_ms"]
    print("abl %3s" % "$a", "step %.3f" % d["ms_per_step"], " ".join("%s %.3f" % (n, k[n]) for n in ("down3", "down4", "down5", "down6", "up1", "up2", "up3", "up4", "up5")))
except Exception as e:
    print("abl $a failed", e)
PY
done
