#!/bin/bash
# Copy the round's records out of gpurun_out/<tag>{,_f16}/ (scripts/gpu_r06.sh <tag> tests bench extras prof proff16 lat) into profiles/<tag>_* and regenerate the summaries:
#   bash scripts/collect_round.sh r06
set -eu
TAG=${1:-r06}; G=gpurun_out/$TAG
cp $G/bench.json profiles/${TAG}_bench_n1.json
for m in f16 f16x2 c5_f16; do [ -f $G/bench_$m.json ] && cp $G/bench_$m.json profiles/${TAG}_bench_$m.json; done
[ -f $G/bench_native.json ] && cp $G/bench_native.json profiles/${TAG}_bench_native_n1.json
[ -f $G/bench_plain_same_box.json ] && cp $G/bench_plain_same_box.json profiles/${TAG}_bench_plain_same_box_as_native.json
[ -f $G/c4.json ] && cp $G/c4.json profiles/${TAG}_c4.json
[ -f $G/latency_c2.json ] && cp $G/latency_c2.json profiles/${TAG}_latency_c2.json
[ -f gpurun_out/${TAG}_latency.json ] && cp gpurun_out/${TAG}_latency.json profiles/${TAG}_latency.json
python scripts/summarize_profiles.py $TAG
python scripts/summarize_profiles.py ${TAG}_f16
for f in spleeterrt_amd/csrc/*.hip; do python scripts/kernel_resources.py $f; done > profiles/${TAG}_kernel_resources.txt 2>/dev/null
python scripts/state_at_head.py $TAG > profiles/${TAG}_state_at_head.md
python scripts/state_at_head.py $TAG f16 > profiles/${TAG}_state_at_head_f16.md
tail -22 profiles/${TAG}_state_at_head_f16.md
