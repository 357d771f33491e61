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


rows = list(csv.DictReader(open(os.path.join(src, "trace", "t_kernel_stats.csv"))))
with open(os.path.join("profiles", tag + "_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        if "srt_" in r["Name"]:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])


def pmc(sub):
    p = os.path.join(src, sub, "p_counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    if not os.path.exists(p):
        return agg, cnt, dur
    for r in csv.DictReader(open(p)):
        if "srt_" not in r["Kernel_Name"]:
            continue
        k = short(r["Kernel_Name"])
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
