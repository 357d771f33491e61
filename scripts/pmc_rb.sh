set -u
R=$PWD; OUT=$R/gpurun_out/r05_rb; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "more_than_four" 2>&1 | tail -3
export SPLEETERRT_LIB=$R/spleeterrt_amd/libspleeterrt_amd_tuning.so SRT_BENCH_NOCHECK=1
cd /tmp
for t in default encrb=1 decrb=0; do
  if [ $t = default ]; then unset SRT_TUNE; else export SRT_TUNE=$t; fi
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_$t -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$t.err
  python - <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open("$OUT/pmc_$t/p_counter_collection.csv")):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "wino" not in k or "pack" in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in sorted(agg): print("$t", k, {c: round(v / len(n[k]) / 1e6, 2) for c, v in agg[k].items()})
PY
done
