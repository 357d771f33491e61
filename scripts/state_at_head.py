#!/usr/bin/env python3
"""One-page kernel table "as of HEAD" from the round's committed records:
    python scripts/state_at_head.py r06 > profiles/r06_state_at_head.md            # the headline mode (fp32, 4 stems)
    python scripts/state_at_head.py r06 f16 > profiles/r06_state_at_head_f16.md    # BASELINE configs[4]: 5 stems, fp16-MFMA conv, fp16 activation storage
Inputs: profiles/<tag>_bench_n1.json | <tag>_bench_c5_f16.json (bench.py line: per-layer kernel, ms, executed / HBM fractions), profiles/<tag>_pmc.json |
<tag>_f16_pmc.json (counter passes: HBM bytes per launch, matrix-pipe busy, wait fraction, LDS bank conflicts), profiles/<tag>_kernel_resources.txt (VGPR / LDS /
spills from the code objects)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
mode = sys.argv[2] if len(sys.argv) > 2 else "f32"
P = os.path.join(ROOT, "profiles")
bench_name = tag + ("_bench_n1.json" if mode == "f32" else "_bench_c5_f16.json")
pmc_name = tag + ("_pmc.json" if mode == "f32" else "_f16_pmc.json")
b = json.load(open(os.path.join(P, bench_name)))
pmc = json.load(open(os.path.join(P, pmc_name))) if os.path.exists(os.path.join(P, pmc_name)) else {}
prec = b["config"]["precision"]
act16 = prec == "f16" and bench.F % 256 == 0
res = {}
rp = os.path.join(P, tag + "_kernel_resources.txt")
if os.path.exists(rp):
    for ln in open(rp):
        m = re.match(r"(\S.*?)\s+vgpr\s+(\d+)\s+agpr\s+(\d+)\s+sgpr\s+(\d+)\s+scratch\s+(\d+)\s+occ\s+(\d+)\s+lds\s+(\d+)\s+spill v(\d+) s(\d+)", ln)
        if m:
            res[m.group(1).strip()] = dict(vgpr=int(m.group(2)), lds=int(m.group(7)), occ=int(m.group(6)), sv=int(m.group(8)), ss=int(m.group(9)))


def find(table, sym):
    if sym in table:
        return table[sym]
    return next((v for k, v in table.items() if bench.same_kernel(k, sym)), None)


inst = b["config"]["stems"] * b["config"]["tiles_per_gpu"]
rows_total = b["config"]["tiles_per_gpu"] * bench.T
order = ["stft", "down1", "down2", "down3", "down4", "down5", "down6", "up1", "up2", "up3", "up4", "up5", "up6", "up7", "istft"]
masks16 = b["layer_kernels"].get("up7", "").replace(" ", "").endswith(",4,true>")     # fp16 mode: the engine's own mask buffer holds halves (bench.py counts the same)
dsp_kb = {"stft": 48.8, "istft": 32.8 + b["config"]["stems"] * (12.0 if masks16 else 16.0)}
print("Kernel table as of HEAD (%s; `%s`, %d tiles x %d stems of %dx%d, %s): step %.3f ms = %.0f x real-time, %.3f M frames/s.\n" % (
    tag, "profiles/%s" % bench_name, b["config"]["tiles_per_gpu"], b["config"]["stems"], bench.T, bench.F, b["dtype"], b["ms_per_step"], b["value"], b["frames_per_s"] / 1e6))
print("| layer | kernel | ms | bound | executed frac of MFMA peak | algorithmic HBM frac | counter traffic / algorithmic bytes | MFMA busy | wait frac | LDS conflict cycles | VGPR / LDS KB / WG per CU-ish occ | spills (v / s) |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
groups = {}
for name in order:
    if name in b["kernel_ms"] and name in b["layer_kernels"]:
        groups.setdefault(b["layer_kernels"][name].split(" + ")[0], []).append(name)
for name in order:
    if name not in b["kernel_ms"]:
        continue
    sym = b["layer_kernels"].get(name, "srt_%s_kernel" % name if name == "stft" else "srt_istft_ola3_kernel<4>")
    first = sym.split(" + ")[0]
    pm, rs = find(pmc, first), find(res, first)
    ms = b["kernel_ms"][name]
    if name in dsp_kb:
        alg = dsp_kb[name] * 1024.0 * rows_total
        ex, hb, bound = "", "%.2f" % (alg / (ms * 1e-3) / 8e12), "HBM"
    else:
        alg = bench.layer_bytes(name, prec, act16, b["config"]["stems"], masks16) * inst
        ex, hb = "%.2f" % b["layer_executed_frac"][name], "%.2f" % b["layer_hbm_frac"][name]
        bound = "HBM" if b["layer_hbm_frac"][name] > b["layer_executed_frac"][name] else "MFMA"
    tr = busy = wait = confl = ""
    if pm:
        if "hbm_read_bytes_per_launch" in pm and "hbm_write_bytes_per_launch" in pm:
            peers = groups.get(first, [name])                                # the counters are per kernel SYMBOL: averaged over the layers it ran
            alg_avg = alg if name in dsp_kb else sum(bench.layer_bytes(k, prec, act16, b["config"]["stems"], masks16) for k in peers) * inst / len(peers)
            tr = "%.2f%s" % ((pm["hbm_read_bytes_per_launch"] + pm["hbm_write_bytes_per_launch"]) / alg_avg, " (avg of %s)" % "-".join([peers[0], peers[-1]]) if len(peers) > 1 else "")
        busy = "%.2f" % pm["mfma_busy_frac"] if pm.get("mfma_busy_frac") else ""
        wait = "%.2f" % pm["wait_any_frac"] if "wait_any_frac" in pm else ""
        confl = "%.1f M" % (pm["sq"]["SQ_LDS_BANK_CONFLICT"] / 1e6) if "sq" in pm and "SQ_LDS_BANK_CONFLICT" in pm["sq"] else ""
    rtxt = "%d / %.1f / %d" % (rs["vgpr"], rs["lds"] / 1024.0, rs["occ"]) if rs else ""
    sp = "%d / %d" % (rs["sv"], rs["ss"]) if rs else ""
    print("| %s | `%s` | %.3f | %s | %s | %s | %s | %s | %s | %s | %s | %s |" % (name, sym, ms, bound, ex, hb, tr, busy, wait, confl, rtxt, sp))
rf = b["roofline"]
print("\nDominant kernel `%s` (%s): %.1f executed TFLOP/s = **%.3f** of the %s-MFMA peak (algorithmic %.1f TFLOP/s); step %.3f of the MFMA peak executed, "
      "**%.3f** of the HBM peak on algorithmic bytes (%.2f GB per step).  `occ` = waves per SIMD from the code object (a 512-thread workgroup = 2).  Counter columns: separate "
      "`rocprofv3 --pmc` passes of the same command (`profiles/%s`); per launch, traffic = FETCH_SIZE x 2 + WRITE_SIZE." % (
          rf["kernel"], ", ".join(rf["layers"]), rf.get("mfma", rf)["achieved"] if rf["bound"] == "hbm" else rf["achieved"],
          rf.get("mfma", rf)["frac"] if rf["bound"] == "hbm" else rf["frac"], "fp16" if bench.is_f16_kernel(rf["kernel"]) else "fp32", rf["algorithmic_tflops"], rf["step"]["frac"],
          rf["step"]["hbm"]["frac"], rf["step"]["hbm"]["algorithmic_bytes"] / 1e9, pmc_name))
