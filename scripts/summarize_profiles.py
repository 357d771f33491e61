#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (scripts/profile_gpu.sh) into the small tracked summaries under profiles/:
   profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats (our kernels only)
   profiles/<tag>_pmc.json           per-kernel PMC sums: SQ activity, HBM bytes per launch (corrected), effective clock
HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1 KB per count and, on
gfx950, FETCH_SIZE under-reports wide coalesced reads by exactly 2x, so read bytes = 2 * FETCH_SIZE * 1024 (upper bound
for narrow accesses); WRITE_SIZE is taken at face value (uncalibrated per the guide)."""
import collections
import csv
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)


def short(n):
    return re.sub(r"^void |\((const )?Srt\w+Params.*$", "", n).strip()


# The engine's first launch of a C8 layer shape TIMES six workgroup-count candidates (c8_tuned, csrc/srt_nn5.hip): those launches carry the layer's kernel symbol and
# would pull every per-launch average away from the step's.  The summaries therefore keep, per symbol, only the dispatches of the TIMED passes: the last
# (layers of one step that run the symbol) x (passes) of the process - 5 passes for the kernel trace, 1 for each counter pass (scripts/profile_gpu*.sh).  The layer ->
# symbol map is the one the traced run printed (bench_under_trace.json); symbols it does not name (weight packing, conversions) keep every dispatch.
per_step = collections.Counter()
try:
    lk = json.load(open(os.path.join(src, "bench_under_trace.json")))["layer_kernels"]
    for v in lk.values():
        for sym in v.split(" + "):
            per_step[sym.strip()] += 1
except Exception as ex:
    print("no bench_under_trace.json (%s): every dispatch counts" % ex)
TRACE_PASSES, PMC_PASSES = 5, 1


def timed(disp, k, passes):
    """disp: list of (dispatch id, ...) of symbol k in any order -> the timed passes' share"""
    disp = sorted(disp, key=lambda t: t[0])
    n = per_step.get(k, 0) * passes
    return disp[-n:] if 0 < n < len(disp) else disp


raw = list(csv.DictReader(open(os.path.join(src, "trace", "t_kernel_stats.csv"))))
with open(os.path.join("profiles", tag + "_kernel_stats_raw.csv"), "w") as f:      # rocprofv3 --stats as it came (every dispatch of the process, tuning and warm-up included)
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in raw:
        if "srt_" in r["Name"]:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
byk = collections.defaultdict(list)
for r in csv.DictReader(open(os.path.join(src, "trace", "t_kernel_trace.csv"))):
    if "srt_" in r["Kernel_Name"]:
        byk[short(r["Kernel_Name"])].append((int(r["Dispatch_Id"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
stat = {}
for k, disp in byk.items():
    d = [x[1] for x in timed(disp, k, TRACE_PASSES)]
    stat[k] = (len(d), sum(d), sum(d) / len(d), min(d), max(d), len(disp))
tot = sum(v[1] for v in stat.values())
with open(os.path.join("profiles", tag + "_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "DispatchesInProcess"])
    for k, v in sorted(stat.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, v[0], v[1], "%.6f" % v[2], "%.2f" % (100.0 * v[1] / tot), v[3], v[4], v[5]])


def pmc(sub):
    p = os.path.join(src, sub, "p_counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    if not os.path.exists(p):
        return agg, cnt, dur
    allrows = [r for r in csv.DictReader(open(p)) if "srt_" in r["Kernel_Name"]]
    ids = collections.defaultdict(set)
    for r in allrows:
        ids[short(r["Kernel_Name"])].add(int(r["Dispatch_Id"]))
    keep = {k: set(t[0] for t in timed([(i,) for i in v], k, PMC_PASSES)) for k, v in ids.items()}
    for r in allrows:
        k = short(r["Kernel_Name"])
        if int(r["Dispatch_Id"]) not in keep[k]:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in cnt[k]:
            cnt[k].add(r["Dispatch_Id"])
            dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return agg, cnt, dur


out = {}
sq, sqn, sqd = pmc("pmc_sq")
fe, fen, _ = pmc("pmc_fetch")
wr, wrn, _ = pmc("pmc_write")
ck, ckn, ckd = pmc("pmc_clk")
for k in sorted(set(sq) | set(fe) | set(wr) | set(ck)):
    d = {}
    if k in sq:
        n = len(sqn[k]); v = sq[k]
        d["launches_profiled"] = n
        d["sq"] = {c: v[c] / n for c in v}
        d["wait_any_frac"] = v["SQ_WAIT_ANY"] / max(v["SQ_WAVE_CYCLES"], 1)
        d["active_inst_frac"] = v["SQ_ACTIVE_INST_ANY"] / max(v["SQ_WAVE_CYCLES"], 1)
    if k in fe:
        d["hbm_read_bytes_per_launch"] = 2.0 * fe[k]["FETCH_SIZE"] * 1024 / len(fen[k])
    if k in wr:
        d["hbm_write_bytes_per_launch"] = wr[k]["WRITE_SIZE"] * 1024 / len(wrn[k])
    if k in ck and ckd[k] > 0:
        d["effective_clock_ghz"] = ck[k]["GRBM_GUI_ACTIVE"] / 8.0 / ckd[k]       # counter is summed over the 8 XCDs
    if k in sq and sqd[k] > 0:
        d["sq_pass_avg_ns"] = sqd[k] / len(sqn[k])
        if "effective_clock_ghz" in d and sq[k].get("SQ_VALU_MFMA_BUSY_CYCLES"):
            # matrix-pipe busy fraction at the clock the chip held: busy cycles are summed over the 1024 SIMDs
            # (64 per v_mfma_f32_32x32x2_f32); 1.0 = every SIMD's matrix pipe busy for the whole launch
            d["mfma_busy_frac"] = sq[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * d["effective_clock_ghz"] * sqd[k])
    out[k] = d
json.dump(out, open(os.path.join("profiles", tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
print("wrote profiles/%s_kernel_stats.csv and profiles/%s_pmc.json (%d kernels)" % (tag, tag, len(out)))
