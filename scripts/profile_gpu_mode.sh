#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash scripts/profile_gpu_mode.sh <tag> <precision> <stems>
# The same passes as scripts/profile_gpu.sh (kernel trace + stats; separate --pmc passes: SQ activity, FETCH_SIZE, WRITE_SIZE, clocks - never
# combined with sys/hip/hsa tracing) for a labelled non-headline mode, e.g. BASELINE configs[4]:  r06_f16 f16 5
# Output: gpurun_out/<tag>/{trace,pmc_sq,pmc_fetch,pmc_write,pmc_clk}; scripts/summarize_profiles.py <tag> turns it into profiles/<tag>_*.
set -u
TAG=${1:-r06_f16}; PREC=${2:-f16}; STEMS=${3:-5}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--no-cpu-baseline --precision $PREC --stems $STEMS"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py --steps 5 --warmup 2 $ARGS > $OUT/bench_under_trace.json 2> $OUT/trace.err
B1="python $R/bench.py --steps 1 --warmup 1 $ARGS"
MOPS=SQ_INSTS_VALU_MFMA_MOPS_F16
[ "$PREC" = f32 ] && MOPS=SQ_INSTS_VALU_MFMA_MOPS_F32
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES $MOPS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
    --kernel-trace --output-format csv -d $OUT/pmc_sq -o p -- $B1 > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o p -- $B1 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o p -- $B1 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_clk -o p -- $B1 > /dev/null 2> $OUT/pmc_clk.err
cd $R
find $OUT -name "*.csv" | head -30
