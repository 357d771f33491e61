#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash scripts/profile_gpu.sh <tag>
# Produces gpurun_out/<tag>/{trace,pmc_*}: rocprofv3 kernel trace + stats of the default bench command, and separate
# PMC passes (never combined with sys/hip/hsa tracing): SQ activity, HBM read bytes, HBM write bytes, clocks.
set -u
TAG=${1:-r02}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BENCH > $OUT/bench_under_trace.json 2> $OUT/trace.err
B1="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
    --kernel-trace --output-format csv -d $OUT/pmc_sq -o p -- $B1 > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o p -- $B1 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o p -- $B1 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_clk -o p -- $B1 > /dev/null 2> $OUT/pmc_clk.err
cd $R
find $OUT -name "*.csv" | head -30
