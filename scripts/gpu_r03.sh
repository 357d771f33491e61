#!/bin/bash
# One gpurun call of round 3:  bash scripts/gpu_r03.sh <tag> <stages...>
#   ubench   scripts/ubench/* microbenchmarks        tests    pytest -m gpu            bench    headline bench line
#   newtests only the tests added in round 3         prof     scripts/profile_gpu.sh   extras   f16 benches, c4 stream, latency script
set -u
TAG=${1:-r03}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for st in "$@"; do
  case $st in
    ubench) for b in scripts/ubench/*; do [ -x "$b" ] && [ ! -d "$b" ] && { echo "== $b"; timeout 120 "$b"; } ; done > $OUT/ubench.log 2>&1; cat $OUT/ubench.log ;;
    tests) ( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=12 ) > $OUT/pytest.log 2>&1; tail -25 $OUT/pytest.log ;;
    newtests) ( time timeout 1200 python -m pytest tests/test_rccl.py tests/test_latency.py tests/test_gpu_parity.py tests/test_dropin_host.py -m gpu -q --durations=12 -s \
        -k "odd_geometries or winograd or batch_invariant or any_length or rccl or nccl or latency or chunks or forward_layers or fp16_mfma" ) > $OUT/newtests.log 2>&1; grep -v "^$" $OUT/newtests.log | tail -60 ;;
    bench) timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json; tail -3 $OUT/bench.err ;;
    prof) timeout 900 bash scripts/profile_gpu.sh $TAG > $OUT/profile.log 2>&1; tail -3 $OUT/profile.log ;;
    extras)
      for prec in f16 f16x2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --precision $prec > $OUT/bench_$prec.json 2>> $OUT/bench.err; done
      timeout 300 python scripts/stream_c4.py --repeats 3 --out $OUT/c4.json > $OUT/c4.log 2>&1; tail -c 400 $OUT/c4.log
      timeout 200 python scripts/latency_c2.py f32 --out $OUT/latency_c2.json > $OUT/lat.log 2>&1; grep "C2 latency" $OUT/lat.log ;;
  esac
done
