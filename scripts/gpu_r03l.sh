#!/bin/bash
# Column walk of the Winograd units (wino_sp_xy): parity, timing against winowalk=0, and one FETCH_SIZE pass of each.
set -u
TAG=${1:-r03l}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "winograd or layer_taps or forward_layers or odd" 2>&1 | tail -5 | tee $OUT/parity.log
bash scripts/gpu_tune.sh $TAG f32 "SRT_TUNE=winowalk=0"
cd /tmp
B1="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_walk -o p -- $B1 > /dev/null 2> $OUT/pmc_fetch_walk.err
SPLEETERRT_LIB=$R/spleeterrt_amd/libspleeterrt_amd_tuning.so SRT_TUNE=winowalk=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_row -o p -- $B1 > /dev/null 2> $OUT/pmc_fetch_row.err
cd $R
python - <<PY
import csv, glob, collections
for tag in ("walk", "row"):
    f = glob.glob("$OUT/pmc_fetch_%s/**/*counter_collection.csv" % tag, recursive=True)
    if not f: print(tag, "no csv"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == "FETCH_SIZE": acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        if "wino" in k and "pack" not in k: print(tag.ljust(5), k.ljust(60), "launches", len(v), "read GB/launch (FETCH_SIZE x 2)", round(sum(v) / len(v) * 1024 * 2 / 1e9, 3))
PY
