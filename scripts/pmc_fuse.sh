#!/bin/bash
# Counters of up6 / head as two kernels (SPLEETERRT_FUSE_HEAD=0) and as one pass (=1), both storage modes.  Run ON THE GPU BOX:  bash scripts/pmc_fuse.sh [tag]
# Separate --pmc passes (SQ activity | instruction mix | FETCH_SIZE | WRITE_SIZE), --kernel-trace only; output gpurun_out/<tag>/fuse_pmc.json (copied to profiles/ by hand).
set -u
TAG=${1:-r06_fuse}
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for mode in "f32 4" "f16 5"; do
  set -- $mode; PREC=$1; STEMS=$2
  for fuse in 0 1; do
    export SPLEETERRT_FUSE_HEAD=$fuse
    B1="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --precision $PREC --stems $STEMS"
    D=$OUT/${PREC}_fuse$fuse
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $D/sq -o p -- $B1 > /dev/null 2> $D.sq.err
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $D/mix -o p -- $B1 > /dev/null 2> $D.mix.err
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/fetch -o p -- $B1 > /dev/null 2> $D.fetch.err
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/write -o p -- $B1 > /dev/null 2> $D.write.err
  done
done
cd $R
python - <<PY
import csv, collections, json, os, re
out = {}
for prec in ("f32", "f16"):
    for fuse in (0, 1):
        d = "$OUT/%s_fuse%d" % (prec, fuse)
        res = collections.defaultdict(dict)
        for sub in ("sq", "mix", "fetch", "write"):
            p = os.path.join(d, sub, "p_counter_collection.csv")
            if not os.path.exists(p):
                continue
            agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set); dur = collections.defaultdict(float)
            for r in csv.DictReader(open(p)):
                k = re.sub(r"^void |\(.*$", "", r["Kernel_Name"]).strip()
                if not ("up6" in k or "head" in k):
                    continue
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Dispatch_Id"] not in n[k]:
                    n[k].add(r["Dispatch_Id"]); dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            for k in agg:
                for c, v in agg[k].items():
                    res[k][c] = v / len(n[k])
                res[k]["launch_us_" + sub] = dur[k] / len(n[k]) / 1e3
        for k, v in res.items():
            if "FETCH_SIZE" in v: v["read_GB"] = v["FETCH_SIZE"] * 2 * 1024 / 1e9           # gfx950 correction (MI355X_MICROARCH.md)
            if "WRITE_SIZE" in v: v["write_GB"] = v["WRITE_SIZE"] * 1024 / 1e9
        out["%s fuse=%d" % (prec, fuse)] = res
json.dump(out, open("$OUT/fuse_pmc.json", "w"), indent=1)
for k, v in out.items():
    for kk, c in v.items():
        print(k, kk, {a: (round(b / 1e6, 2) if b > 1e5 else round(b, 3)) for a, b in sorted(c.items())})
PY
