#!/bin/bash
# One gpurun call of round 4:  bash scripts/gpu_r04.sh <tag> <stages...>
#   tests    pytest -m gpu (whole suite)       newtests  only the tests added in round 4          bench   headline bench line
#   prof     scripts/profile_gpu.sh            extras    f16 benches, c4 stream, latency script    quick   the short parity tests of the conv kernels
#   proff16  scripts/profile_gpu_f16.sh (HBM counters of the fp16-storage mode)             benchq  short bench line without the CPU baseline
set -u
TAG=${1:-r04}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for st in "$@"; do
  case $st in
    tests) ( time timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 ) > $OUT/pytest.log 2>&1; tail -30 $OUT/pytest.log ;;
    newtests) ( time timeout 1500 python -m pytest tests/test_multi_device.py tests/test_gpu_parity.py tests/test_rccl.py tests/test_latency.py -m gpu -q --durations=12 -s \
        -k "multi or several_gpu or shipped_launch or reference_default or up6_and_head or full_size_batch or fp16_modes or latency" ) > $OUT/newtests.log 2>&1; grep -v "^$" $OUT/newtests.log | tail -60 ;;
    quick) ( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --durations=8 -k "odd_geometries or winograd_decoder_layers or up6_streamed or forward_layers or stft_matches or istft_roundtrip or separate_end_to_end" ) > $OUT/quick.log 2>&1; tail -15 $OUT/quick.log ;;
    bench) timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 1800 $OUT/bench.json; tail -3 $OUT/bench.err ;;
    benchq) timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/benchq.json 2> $OUT/bench.err; python - <<PY
import json
d = json.load(open("$OUT/benchq.json")); print("ms/step", round(d["ms_per_step"], 3), d["kernel_ms"]); print(d["layer_executed_frac"])
PY
      tail -3 $OUT/bench.err ;;
    prof) timeout 900 bash scripts/profile_gpu.sh $TAG > $OUT/profile.log 2>&1; tail -3 $OUT/profile.log ;;
    proff16) timeout 600 bash scripts/profile_gpu_f16.sh $TAG > $OUT/profile_f16.log 2>&1; tail -2 $OUT/profile_f16.log ;;
    extras)
      for prec in f16 f16x2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --precision $prec > $OUT/bench_$prec.json 2>> $OUT/bench.err; done
      timeout 300 python scripts/stream_c4.py --repeats 3 --out $OUT/c4.json > $OUT/c4.log 2>&1; tail -c 400 $OUT/c4.log
      timeout 200 python scripts/latency_c2.py f32 --out $OUT/latency_c2.json > $OUT/lat.log 2>&1; grep "C2 latency" $OUT/lat.log ;;
  esac
done
