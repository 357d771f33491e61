#!/bin/bash
# HBM traffic of the fp16-MFMA mode (fp16 activation storage): rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes of
# `bench.py --precision f16` (separate --pmc passes, never combined with sys tracing).  Output: gpurun_out/<tag>_f16/
set -u
TAG=${1:-r02}_f16
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --precision f16"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $B > $OUT/bench_under_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o p -- $B > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o p -- $B > /dev/null 2> $OUT/pmc_write.err
cd $R
find $OUT -name "*.csv" | head
