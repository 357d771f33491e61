#!/bin/bash
# One gpurun call: GPU parity suite, headline bench, the configs[3] stream, the configs[1] latency.  Run from the repo root:
#   gpurun --timeout 1500 -- 'bash scripts/gpu_round.sh r02a'
set -u
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1100 python -m pytest tests -m gpu -q --durations=15 ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -c 1500 $OUT/bench.json
timeout 300 python scripts/stream_c4.py --out $OUT/c4.json > $OUT/c4.log 2>&1
tail -c 1200 $OUT/c4.log
timeout 120 python scripts/latency_c2.py > $OUT/lat_c2.log 2>&1
tail -3 $OUT/lat_c2.log
