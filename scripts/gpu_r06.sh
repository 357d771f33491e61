#!/bin/bash
# One gpurun call of round 6:  bash scripts/gpu_r06.sh <tag> <stages...>
#   tests    pytest -m gpu (whole suite)          quick     the layer / launch-shape parity tests        bench    headline bench line (with CPU baseline)
#   benchq   short bench line                     ab:<env;env>  benchq under each env assignment         prof     scripts/profile_gpu.sh (4-stem fp32)
#   proff16  scripts/profile_gpu_mode.sh <tag>_f16 f16 5 (BASELINE configs[4])                           c5       the 5-stem f16 line     f16   4-stem f16 line
#   extras   f16 / f16x2 / 5-stem lines, c4 stream, latency script          two      scripts/two_on_one.py        k:<expr>  pytest -k <expr>
#   lat      tests/test_latency.py + latency record                          tune:<prec>:<env;env>  scripts/gpu_tune.sh on the tuning library
set -u
TAG=${1:-r06}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
line() {   # label, env assignment (may be empty), extra bench args
  env $2 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $3 > "$OUT/bench_$1.json" 2>> $OUT/bench.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$1.json")); print("$1".ljust(34), round(d["ms_per_step"], 3), d["kernel_ms"])
except Exception as e:
    print("$1", "failed", e)
PY
}
for st in "$@"; do
  case $st in
    tests) ( time timeout 3000 python -m pytest tests -m gpu -q -x --durations=15 ) > $OUT/pytest.log 2>&1; tail -30 $OUT/pytest.log ;;
    quick) ( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --durations=8 -k "odd_geometries or winograd_decoder_layers or up6_streamed or forward_layers or stft_matches or istft_roundtrip or separate_end_to_end or shipped_launch" ) > $OUT/quick.log 2>&1; tail -15 $OUT/quick.log ;;
    k:*) ( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 -k "${st#k:}" ) > $OUT/k.log 2>&1; tail -15 $OUT/k.log ;;
    bench) timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json; tail -3 $OUT/bench.err ;;
    benchq) line default "" "" ;;
    c5) line c5_f16 "" "--precision f16 --stems 5 --steps 20 --warmup 5" ;;
    f16) line f16 "" "--precision f16 --steps 20 --warmup 5" ;;
    ab:*) IFS=';' read -ra SETS <<< "${st#ab:}"; for t in "${SETS[@]}"; do line "$t" "$t" ""; done ;;
    ab5:*) IFS=';' read -ra SETS <<< "${st#ab5:}"; for t in "${SETS[@]}"; do line "c5_$t" "$t" "--precision f16 --stems 5"; done ;;
    prof) timeout 900 bash scripts/profile_gpu.sh $TAG > $OUT/profile.log 2>&1; tail -3 $OUT/profile.log ;;
    proff16) timeout 900 bash scripts/profile_gpu_mode.sh ${TAG}_f16 f16 5 > $OUT/profile_f16.log 2>&1; tail -2 $OUT/profile_f16.log ;;
    two) timeout 600 python scripts/two_on_one.py --out $OUT/two_on_one.json > $OUT/two.log 2>&1; tail -8 $OUT/two.log ;;
    two16) timeout 600 python scripts/two_on_one.py --precision f16 --stems 5 --out $OUT/two_on_one_f16.json > $OUT/two16.log 2>&1; tail -8 $OUT/two16.log ;;
    lat) ( time timeout 900 python -m pytest tests/test_latency.py -m gpu -q -x ) > $OUT/lat_test.log 2>&1; tail -8 $OUT/lat_test.log ;;
    tune:*) IFS=':' read -r _ PREC SETS <<< "$st"; bash scripts/gpu_tune.sh $TAG/tune $PREC "$SETS" 2>&1 | tail -40 ;;
    extras)
      for prec in f16 f16x2; do line $prec "" "--precision $prec --steps 20 --warmup 5"; done
      line c5_f16 "" "--precision f16 --stems 5 --steps 20 --warmup 5"
      timeout 300 python scripts/stream_c4.py --repeats 3 --out $OUT/c4.json > $OUT/c4.log 2>&1; tail -c 400 $OUT/c4.log
      timeout 200 python scripts/latency_c2.py f32 --out $OUT/latency_c2.json > $OUT/lat.log 2>&1; grep "C2 latency" $OUT/lat.log ;;
  esac
done
