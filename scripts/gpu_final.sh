#!/bin/bash
# One gpurun call for the round's tracked evidence: parity suite, headline bench (+ f16 / f16x2), configs[3] stream, latency,
# rocprofv3 kernel trace + PMC passes.  bash scripts/gpu_final.sh <tag>
set -u
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1100 python -m pytest tests -m gpu -q --durations=8 ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
for prec in f16 f16x2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --precision $prec > $OUT/bench_$prec.json 2>> $OUT/bench.err
done
timeout 300 python scripts/stream_c4.py --repeats 3 --out $OUT/c4.json > $OUT/c4.log 2>&1
tail -c 400 $OUT/c4.log
timeout 200 python scripts/latency_c2.py f32 --out $OUT/latency.json > $OUT/lat.log 2>&1
timeout 200 python scripts/latency_c2.py f16 --out $OUT/latency_f16.json >> $OUT/lat.log 2>&1
grep "C2 latency" $OUT/lat.log
timeout 900 bash scripts/profile_gpu.sh $TAG > $OUT/profile.log 2>&1
tail -3 $OUT/profile.log
