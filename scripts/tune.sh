#!/bin/bash
# run on the GPU box (library built with SRT_TUNING=1): prints per-layer ms for each SRT_TUNE setting
for t in "" "down2=1,up5=1,up4=1" "down2=2,up5=2,up4=2" "down2=3,up5=3,up4=3" "up5=4"; do
  SRT_TUNE="$t" python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('$t'.ljust(24), round(d['ms_per_step'],3), 'down2', k['down2'], 'up4', k['up4'], 'up5', k['up5'])"
done
