#!/usr/bin/env python3
"""bench.py — whole-job throughput of the separation hot path on N MI355X (one process per GPU).

A "step" = one pass of PCM -> STFT -> |.| -> 4 U-Nets -> mask -> iSTFT over one batch of 64 spectrogram tiles
(256 frames x 1024 bins, 4 stems, 44.1 kHz stereo, fp32): BASELINE.json configs[2], the 1-GPU configuration the
metric is quoted on.  Inputs (PCM) and outputs (stems) are resident in HBM.  Multi-GPU is weak scaling: tiles are
independent (main.c:455-495), every rank processes its own 64-tile batch, and the only collective is the RCCL
broadcast of the weight blobs at start-up.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N                 # no launcher in the environment: bench.py starts the N ranks itself (same command as above)
    python bench.py --gpus N --host native   # ONE process: the C host (csrc/srt_multi.hip, main.c:544-673's shape) drives N devices

`--gpus N` is binding: the line is printed only when N ranks (or N native workers) on N distinct devices really ran; with fewer devices
visible the program exits non-zero with a message instead of printing an `n_gpus: 1` line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F, T, STEMS, TILES = 1024, 256, 4, 64
FS, HOP = 44100.0, 1024
FLOP_PER_PIXEL = 23264                      # per T-F pixel per sub-net, SURVEY §8d (sum of 2MNK over the 13 GEMMs)
PEAK_F32_MFMA_TFLOPS = 157.3                # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0               # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA (v_mfma_f32_32x32x16_f16); the 2:1-sparsity figure is never used
PEAK_HBM_TBS = 8.0
# FLOP per launch and per instance (one tile of one stem) for every layer kernel, T=256 F=1024
LAYER_FLOP = {}
_enc = [(2, 16), (16, 32), (32, 64), (64, 128), (128, 256), (256, 512)]
_dec = [(512, 256), (512, 128), (256, 64), (128, 32), (64, 16), (32, 1)]
for i, (ci, co) in enumerate(_enc):
    LAYER_FLOP["down%d" % (i + 1)] = 2 * co * ci * 25 * (T >> (i + 1)) * (F >> (i + 1))
for i, (ci, co) in enumerate(_dec):
    LAYER_FLOP["up%d" % (i + 1)] = 2 * co * ci * 25 * (T >> (6 - i)) * (F >> (6 - i))
LAYER_FLOP["up7"] = 2 * 2 * 16 * T * F
assert sum(LAYER_FLOP.values()) == FLOP_PER_PIXEL * T * F
# Which kernel ran each layer is NOT assumed here: the engine reports the symbol of every timed launch (srtGetTimingKernels).
# Layers that ran in Winograd form (csrc/srt_nn4.hip) EXECUTE 49 MFMA products per 2x2 input block where the layer's algorithm
# (LAYER_FLOP, the reference's direct transposed convolution) has 100: the roofline fraction is computed on the EXECUTED FLOPs (it can
# never exceed 1); the algorithmic rate and the 100/49 speed-up are reported beside it.
WINO_EXECUTED_FRACTION = 0.49


def same_kernel(a, b):
    """kernel symbols that differ only by trailing template arguments (mode flags appended as the code grows, at their defaults in the
    older name): same base name and one argument list is a prefix of the other"""
    def split(sym):
        base, _, args = sym.partition("<")
        return base.strip(), [x.strip() for x in args.rstrip("> ").split(",")] if args else []
    (ba, aa), (bb, ab) = split(a), split(b)
    n = min(len(aa), len(ab))
    return ba == bb and aa[:n] == ab[:n]


def is_f16_kernel(symbol):
    """kernels of csrc/srt_nn3.hip (v_mfma_f32_32x32x16_f16), and up6 - tiled or streamed - when its inputs are halves (4th template argument)"""
    if "_f16<" in symbol or "_c8<" in symbol:          # csrc/srt_nn3.hip, and the C8-form kernels of csrc/srt_nn5.hip (same MFMA)
        return True
    if symbol.startswith("srt_down1_f16_kernel<"):     # down1 of the fp16 mode (csrc/srt_nn2.hip)
        return True
    if symbol.startswith("srt_up6_kernel<") or symbol.startswith("srt_up6_stream_kernel<"):
        args = symbol.split("<", 1)[1].rstrip("> ").split(",")
        return len(args) > 3 and args[3].strip().startswith("true")
    return False


def executed_fraction(symbol, precision="f32"):
    """MFMA products the kernel executes / products of the layer's algorithm (the reference's direct convolution)"""
    if "wino" in symbol:
        return WINO_EXECUTED_FRACTION
    if symbol.startswith("srt_dec_c8<") and [x.strip() for x in symbol.split("<", 1)[1].rstrip("> ").split(",")][4:5] == ["true"]:   # 5th template argument: CS
        return 15.0 * 32 / (25.0 * 16)      # up5, class-stacked: 15 products of 32 rows (2 x-classes x 16 channels) where the algorithm has 25 of 16 rows
    if symbol.startswith("srt_down1_f16_kernel<"):
        return 80.0 / 50.0                  # k-groups of 8 hold the 5 taps of one (channel, ky): 80 products per output where the algorithm has 50 (half-empty M tiles of odd stem counts not counted)
    if precision == "f16x2" and "_f16<" in symbol:
        return 2.0                          # activations split hi + lo: two MFMAs per tap
    return 1.0


def mfma_peak(symbol):
    return PEAK_F16_MFMA_TFLOPS if is_f16_kernel(symbol) else PEAK_F32_MFMA_TFLOPS


def layer_bytes(name, precision, act16, stems=STEMS, masks16=False):
    """ALGORITHMIC HBM bytes of one layer per instance (tile x stem): its input(s) read once, its output(s) written once, at the element size
    the tensors have in this mode (fp16 storage: raw_i, act_i and up_1..5 are halves; the encoder writes raw AND the act(BN(.)) copy).  Weights
    are read once per launch, not per instance, and are left out (<= 13 MB against GBs)."""
    e = 2 if act16 else 4
    if name.startswith("down"):
        i = int(name[4:]) - 1
        ci, co = _enc[i]
        hin, hout = (T >> i) * (F >> i), (T >> (i + 1)) * (F >> (i + 1))
        rd = ci * hin * (4 if i == 0 else e) / (stems if i == 0 else 1)          # the magnitudes are shared by the stems
        wr = co * hout * e * (2 if (act16 and i < 5) else 1)                      # fp16 storage: raw + act copy
        return rd + wr
    if name.startswith("up") and name != "up7":
        i = int(name[2:]) - 1
        ci, co = _dec[i]
        hin = (T >> (6 - i)) * (F >> (6 - i))
        return ci * hin * e + co * 4 * hin * (4 if i == 5 else e)                 # up6's output (the head's input) stays fp32
    return T * F * 4 + 2 * T * F * (2 if masks16 else 4)                          # head: one plane in, two masks out (halves in the fp16 mode's srtSeparate: the engine's own buffer)


# written by scripts/summarize_profiles.py from separate --pmc passes of this same command (latest round first); per precision
# (file, stems of the profiled launch shape; every profile is of the 64-tile batch)
PMC_SUMMARIES = {"f32": [(os.path.join(ROOT, "profiles", f), 4) for f in ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json", "r02_pmc.json", "r02_direct_pmc.json", "r01_pmc.json")],
                 "f16": [(os.path.join(ROOT, "profiles", "r06_f16_pmc.json"), 5)] +                 # BASELINE configs[4] as written: five stems
                        [(os.path.join(ROOT, "profiles", f), 4) for f in ("r04_f16_pmc.json", "r02_f16_pmc.json")],
                 "f16x2": [(os.path.join(ROOT, "profiles", f), 4) for f in ("r04_f16x2_pmc.json",)]}
N_SIMD = 1024                               # 256 CUs x 4 SIMDs: SQ_VALU_MFMA_BUSY_CYCLES is summed over them


def synth_weights(stem, device):
    """Random-init weights of the reference architecture, fp16-representable, spleeterCoeff field order."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(2024 + stem)
    parts = []

    def u(n, scale, off=0.0):
        parts.append((off + scale * (torch.rand(n, generator=g) - 0.5)).half().float())
    for i, (ci, co) in enumerate(_enc):
        u(25 * ci * co, 2.0 * (6.0 / (25 * ci)) ** 0.5); u(co, 0.02)
        if i < 5:
            u(co, 0.2); u(co, 0.5, 1.0)
    for ci, co in _dec:
        u(25 * ci * co, 2.0 * (6.0 / (25 * ci / 4.0)) ** 0.5); u(co, 0.02); u(co, 0.2); u(co, 0.5, 1.0)
    u(32, 2.0 * (6.0 / 16.0) ** 0.5); u(2, 0.02)
    w = torch.cat(parts)
    assert w.numel() == 9822725
    return w.to(device)


def cpu_baseline(sample_tiles=None):
    """Reference CPU path (oracle/_ref: the reference's own C compiled from /root/reference, CPU_GEMM=1 naive GEMM)
    on a bounded sample: tiles fanned out over threads like processMT (main.c:544-673) — one forward (tile, sub-network) with
    the sequential GEMM per thread, every host core busy — plus the reference stft/istft of the same span."""
    os.environ.setdefault("OMP_NUM_THREADS", "1")          # tile-level parallelism only, as processMT does
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pyoracle as O
    flags = open("/proc/cpuinfo").read()
    flavour = "avx2" if (" avx2 " in flags and " fma " in flags and O.ref_path("avx2")) else "exe"
    kind = "reference"
    model = next((ln.split(":", 1)[1].strip() for ln in flags.splitlines() if ln.startswith("model name")), "unknown")
    nproc = os.cpu_count() or 1
    try:
        nproc_avail = len(os.sched_getaffinity(0))
    except AttributeError:
        nproc_avail = nproc
    if O.ref_path(flavour) is None:
        return {"value": None, "unit": "frames/s", "cores": 0, "nproc": nproc, "cpu_model": model, "kind": "reference", "sample": "oracle/_ref not built"}
    cores = nproc_avail                                     # every host core this process may run on
    # bounded sample (~10-30 s of CPU work): one (tile, sub-network) forward per thread keeps every core busy with a quarter of
    # the tiles a one-tile-per-thread split would need (a forward of the naive GEMM takes ~9 s on one core)
    ntiles = sample_tiles or min(max((cores + STEMS - 1) // STEMS, 16), 256)
    coeffs = [O.synth_coeff(s) for s in range(STEMS)]
    n = ntiles * T * HOP
    L, R = O.synth_audio(n, 777, True)
    st = O.RefSTFT(cores=min(cores, 16), flavour=flavour)   # the reference's STFT worker pool (stftFix.c:379-428); 16 is its practical knee
    t0 = time.time()
    re, im = st.stft(L, R)
    t_stft = time.time() - t0
    mags = [O.magnitude_tile(re, im, j * T, T, F) for j in range(ntiles)]

    def work(js):
        j, s = js
        net = O.RefNet(coeffs[s], F, T, 1, flavour)
        y = net(mags[j])
        net.close()
        return y
    t0 = time.time()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(work, [(j, s) for j in range(ntiles) for s in range(STEMS)]))
    t_nn = time.time() - t0
    t0 = time.time()
    for s in range(STEMS):
        st.istft(re, im)
    t_istft = time.time() - t0
    st.close()
    frames = ntiles * T
    total = t_stft + t_nn + t_istft
    return {"value": frames / total, "unit": "frames/s", "x_realtime": frames * HOP / FS / total, "cores": cores, "nproc": nproc,
            "cpu_model": model, "kind": kind,
            "sample": "%d tiles x %d stems (%d frames): stft %.2fs + nn %.2fs + istft %.2fs; %s build of the reference C "
                      "(naive CPU_GEMM=1 GEMM, no MKL), tiles fanned out over threads as processMT does, one (tile, sub-network) forward per thread" % (ntiles, STEMS, frames, t_stft, t_nn, t_istft, flavour)}


def make_line(a, rec):
    """The one JSON line, from a measurement record (measure_ranks on rank 0, or measure_native)."""
    stems, dt, dt_ev, tim, kern, world, rows = a.stems, rec["dt"], rec["dt_ev"], rec["tim"], rec["kern"], rec["world"], rec["rows"]
    nocheck = os.environ.get("SRT_BENCH_NOCHECK") == "1"
    frames_step = rec["frames_step"]                           # frames that actually get a transform (rows - 3: stftFix.c:378)
    frames_total = frames_step * world * a.steps
    fps = frames_total / dt
    per = {}
    for name, ms in tim:
        per.setdefault(name, []).append(ms)
    # per STEP: a layer that goes out as several launches (down1 of five sub-networks: a stack of four + one) counts with their sum
    # (the divisor is the number of passes the timing log really holds - a saturated log (SRT_TIMING_MAX launches) or a native-host window that is not
    # exactly `steps` passes would otherwise under-report every kernel silently)
    passes = len(per.get("stft", [])) or a.steps
    assert nocheck or passes == a.steps, "timing log holds %d passes of the step, %d were asked for (log saturated?)" % (passes, a.steps)
    avg = {k: float(np.sum(v)) / passes for k, v in per.items()}
    inst = stems * a.tiles
    nn_ms = sum(v for k, v in avg.items() if k in LAYER_FLOP or k == "actcopy")      # actcopy: the fallback bn+act pass in front of the first Winograd-form encoder layer (normally its producer writes the copy)
    nn_flop = FLOP_PER_PIXEL * T * F * inst
    # dominant kernel = the kernel SYMBOL with the largest share of the step (what tops rocprofv3 --stats);
    # achieved = algorithmic FLOPs per launch / average launch duration (HIP events on the engine's stream)
    layer_syms = {}
    for name, sym in kern:
        if sym not in layer_syms.setdefault(name, []):
            layer_syms[name].append(sym)
    layer_kernel = {k: " + ".join(v) for k, v in layer_syms.items()}           # (one symbol per layer, except a layer split into launch groups)
    sym_ms, sym_flop, sym_n = {}, {}, {}
    for k in avg:
        if k in LAYER_FLOP:
            sy = layer_kernel.get(k, k)
            sym_ms[sy] = sym_ms.get(sy, 0.0) + avg[k]
            sym_flop[sy] = sym_flop.get(sy, 0.0) + LAYER_FLOP[k] * inst
            sym_n[sy] = sym_n.get(sy, 0) + 1
    # FLOPs the matrix pipe EXECUTES in one step: the Winograd-form layers issue 0.49 of their algorithmic products
    prec = a.precision
    act16 = prec == "f16" and F % 256 == 0                # fp16 activation storage (csrc/srt_engine.hip: act16)
    masks16 = layer_kernel.get("up7", "").replace(" ", "").endswith(",4,true>")      # the head wrote the engine's own mask buffer as halves (fp16 mode, csrc/srt_engine.hip: separate_issue)
    mask_pcm_kb = 8.0 + (4.0 if masks16 else 8.0)         # per stem and frame: 8 KB of PCM out + 2 x 1024 mask values in
    nn_exec_flop = sum(LAYER_FLOP[k] * inst * executed_fraction(layer_kernel.get(k, k), prec) for k in avg if k in LAYER_FLOP)
    # the step's MFMA time budget: every layer's executed FLOPs at the peak of the MFMA it runs on (fp16 modes mix both pipes)
    nn_peak_ms = sum(LAYER_FLOP[k] * inst * executed_fraction(layer_kernel.get(k, k), prec) / (mfma_peak(layer_kernel.get(k, k)) * 1e12) * 1e3 for k in avg if k in LAYER_FLOP)
    dom = max(sym_ms, key=lambda k: sym_ms[k])
    dom_ms = sym_ms[dom] / sym_n[dom]
    dom_flop = sym_flop[dom] / sym_n[dom]
    dom_alg_tflops = dom_flop / (dom_ms * 1e-3) / 1e12    # algorithmic (the reference's direct convolution)
    dom_exec = executed_fraction(dom, prec)
    dom_tflops = dom_alg_tflops * dom_exec                # what the matrix pipe executes: the roofline figure
    dom_peak = mfma_peak(dom)
    dom_layers = sorted(k for k in avg if layer_kernel.get(k) == dom)
    dom_bytes = sum(layer_bytes(k, prec, act16, stems, masks16) for k in dom_layers) * inst / len(dom_layers)      # algorithmic HBM bytes per launch
    dom_gbs = dom_bytes / (dom_ms * 1e-3) / 1e9
    # traffic / mfma_busy come from the committed counter passes (profiles/), NOT from this run: they are attached only when the profile is of
    # the SAME kernel symbol at the same launch shape and its launch duration agrees with this run's within 10 %; otherwise they are null and
    # `profile_check` says why (a stale profile must never describe a changed kernel)
    traffic = mfma_busy = pmc_file = None
    profile_check = {"status": "no committed counter profile holds this kernel symbol", "run_avg_ms_per_launch": dom_ms}
    for pf, pf_stems in PMC_SUMMARIES[prec]:
        try:
            allpm = json.load(open(pf))
            pm = allpm.get(dom) or next((v for k, v in allpm.items() if same_kernel(k, dom)), None)
            if pm and (a.tiles != TILES or stems != pf_stems):
                # (a profile of another launch shape says nothing about this run's launches: keep looking, and say so if nothing fits)
                profile_check = {"status": "the committed profiles holding this kernel are of another launch shape (%s: %d tiles x %d stems); this run is %d x %d" % (
                    os.path.relpath(pf, ROOT), TILES, pf_stems, a.tiles, stems), "run_avg_ms_per_launch": dom_ms}
                continue
            if pm:
                prof_ms = (pm.get("sq_pass_avg_ns") or pm.get("avg_ns") or 0.0) * 1e-6
                rel = abs(prof_ms - dom_ms) / dom_ms if prof_ms else None
                profile_check = {"file": os.path.relpath(pf, ROOT), "profile_avg_ms_per_launch": prof_ms or None, "run_avg_ms_per_launch": dom_ms,
                                 "relative_difference": rel, "tolerance": 0.10}
                if rel is None or rel > 0.10:
                    profile_check["status"] = "MISMATCH: the profiled kernel's duration differs from this run's by more than 10 % - traffic / mfma_busy withheld"
                    break
                profile_check["status"] = "ok"
                traffic = pm["hbm_read_bytes_per_launch"] + pm["hbm_write_bytes_per_launch"]
                # matrix-pipe busy fraction: SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs; 64 per v_mfma_f32_32x32x2_f32)
                # / (1024 x shader cycles of the launch = GRBM_GUI_ACTIVE / 8 XCDs), both from the profiled passes
                mfma_busy = pm.get("mfma_busy_frac")
                if mfma_busy is None and "sq" in pm and pm.get("effective_clock_ghz") and pm.get("sq_pass_avg_ns"):
                    mfma_busy = pm["sq"]["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * pm["effective_clock_ghz"] * pm["sq_pass_avg_ns"])
                pmc_file = os.path.relpath(pf, ROOT)
                break
        except Exception:
            continue
    step_ms = dt / a.steps * 1e3
    res = {
        "metric": "x_realtime (%d-stem separation, 44.1 kHz stereo, PCM->stems resident in HBM); frames_per_s alongside" % stems,
        "value": fps * HOP / FS, "unit": "x real-time", "frames_per_s": fps,
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": step_ms, "ms_per_step_events": dt_ev / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"f32": "f32", "f16": "f16 products, f32 accumulate (conv only; STFT/iSTFT f32)", "f16x2": "f16x2 split products (exact in f32), f32 accumulate"}[a.precision],
        "data": "synthetic",
        "config": {"workload": "%d-stem, %s, batch=%d spectrogram tiles of %dx%d per GPU (%s); "
                               "%d frames = %.1f s of audio per GPU per step" % (stems, {"f32": "fp32", "f16": "fp16-MFMA conv + fp32 STFT/iSTFT", "f16x2": "fp16 hi+lo MFMA conv + fp32 STFT/iSTFT"}[a.precision],
                                                                                 a.tiles, T, F,
                                                                                 "BASELINE configs[4]: 5 stems, fp16 MFMA conv with fp32 STFT/iSTFT" if (stems == 5 and a.precision == "f16")
                                                                                 else "BASELINE configs[2]" if (stems == 4 and a.precision == "f32") else "BASELINE configs[2]'s batch in a labelled non-headline mode",
                                                                                 frames_step, frames_step * HOP / FS),
                   "stems": stems, "tiles_per_gpu": a.tiles, "T": T, "F": F, "parallelism": "tile-sharded x%d, no data-path collective" % world,
                   "impl": a.impl, "precision": a.precision},
        "host": rec["host"], "distributed": rec["distributed"],
        "roofline": {"bound": "mfma", "kernel": dom, "achieved": dom_tflops, "peak": dom_peak, "unit": "TFLOP/s",
                     "frac": dom_tflops / dom_peak, "traffic": traffic,
                     "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE*2 + WRITE_SIZE, %s)" % pmc_file,
                     "hbm_gbs": (traffic / (dom_ms * 1e-3) / 1e9) if traffic else None,
                     "mfma_busy_frac": mfma_busy, "profile_check": profile_check,
                     "mfma_busy_note": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x shader cycles of the launch at the clock the chip held, GRBM_GUI_ACTIVE/8); frac is against the 2.4 GHz peak",
                     "flop_per_launch": dom_flop * dom_exec, "avg_ms_per_launch": dom_ms, "launches_per_step": sym_n[dom],
                     "layers": sorted(k for k in avg if layer_kernel.get(k) == dom),
                     "kernel_source": "srtGetTimingKernels (the symbol the engine launched in this run)",
                     "note": "achieved / frac / flop_per_launch count the MFMA products the kernel EXECUTES; for a Winograd-form kernel that is 0.49 of the "
                             "layer's algorithmic FLOPs (algorithmic_* below), so frac cannot exceed 1",
                     "algorithmic_flop_per_launch": dom_flop, "algorithmic_tflops": dom_alg_tflops, "algorithmic_speedup": 1.0 / dom_exec,
                     # SURVEY 8(d)'s reading (algorithmic FLOPs of the layer / time / peak): above 1 for a Winograd-form kernel, which is why `frac` is on the executed FLOPs
                     "algorithmic_frac": dom_alg_tflops / dom_peak,
                     # the same kernel against the HBM roofline: ALGORITHMIC bytes per launch (inputs read once + outputs written once at this mode's element sizes) / time
                     "hbm": {"achieved": dom_gbs, "peak": PEAK_HBM_TBS * 1e3, "unit": "GB/s", "frac": dom_gbs / (PEAK_HBM_TBS * 1e3), "algorithmic_bytes_per_launch": dom_bytes,
                             "counter_gbs": (traffic / (dom_ms * 1e-3) / 1e9) if traffic else None},
                     "share_of_step": sym_ms[dom] / (dt_ev / a.steps * 1e3),
                     # the whole path against the MFMA roofline: algorithmic network FLOP of one step / wall time of one step
                     "step": {"achieved": nn_exec_flop / (step_ms * 1e-3) / 1e12, "peak": nn_exec_flop / (nn_peak_ms * 1e-3) / 1e12, "unit": "TFLOP/s",
                              "frac": nn_peak_ms / step_ms,
                              "peak_note": "executed FLOPs of the 13 layers / the time they would take at the dense peak of the MFMA each runs on (157.3 fp32, 2500 fp16)",
                              "algorithmic_tflops": nn_flop / (step_ms * 1e-3) / 1e12, "algorithmic_speedup": nn_flop / nn_exec_flop,
                              "note": "MFMA FLOPs executed by the 13 layers of one step / ms_per_step (STFT, iSTFT and launch gaps included in the time); "
                                      "algorithmic = 23 264 FLOP per T-F pixel per sub-net x pixels of the step (the reference's direct convolutions)"}},
        "nn_stack": {"achieved_tflops": nn_exec_flop / (nn_ms * 1e-3) / 1e12, "frac": nn_peak_ms / nn_ms,
                     "algorithmic_tflops": nn_flop / (nn_ms * 1e-3) / 1e12, "algorithmic_speedup": nn_flop / nn_exec_flop,
                     "ms": nn_ms, "executed_flop": nn_exec_flop, "algorithmic_flop": nn_flop},
        # the HBM-bound stages against the same guide's 8 TB/s: algorithmic bytes per frame / measured kernel time
        "dsp_stages": {name: {"bound": "hbm", "achieved": kb * 1024.0 * rows / (avg[name] * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                              "frac": kb * 1024.0 * rows / (avg[name] * 1e-3) / 8e12, "algorithmic_kb_per_frame": kb}
                       for name, kb in (("stft", 48.8), ("istft", 32.8 + stems * mask_pcm_kb)) if name in avg},
        "kernel_ms": {k: round(v, 4) for k, v in sorted(avg.items())},
        "layer_kernels": {k: layer_kernel[k] for k in sorted(layer_kernel)},
        "layer_tflops": {k: round(LAYER_FLOP[k] * inst / (avg[k] * 1e-3) / 1e12, 2) for k in avg if k in LAYER_FLOP},     # algorithmic
        "layer_executed_frac": {k: round(LAYER_FLOP[k] * inst * executed_fraction(layer_kernel.get(k, k), prec) / (avg[k] * 1e-3) / 1e12 / mfma_peak(layer_kernel.get(k, k)), 4)
                                for k in avg if k in LAYER_FLOP},
        # every layer against the HBM roofline too (algorithmic bytes at this mode's element sizes / kernel time / 8 TB/s)
        "layer_hbm_frac": {k: round(layer_bytes(k, prec, act16, stems, masks16) * inst / (avg[k] * 1e-3) / (PEAK_HBM_TBS * 1e12), 4) for k in avg if k in LAYER_FLOP},
    }
    rf = res["roofline"]
    if rf["hbm"]["frac"] > rf["frac"]:
        # the dominant kernel is nearer the HBM roofline than the MFMA one (the fp16-MFMA mode is HBM-bound): report THAT as the
        # bound, with achieved = algorithmic bytes per launch / average launch duration; the MFMA view stays under "mfma"
        rf["mfma"] = {k: rf[k] for k in ("achieved", "peak", "unit", "frac")}
        rf.update(bound="hbm", achieved=rf["hbm"]["achieved"], peak=rf["hbm"]["peak"], unit="GB/s", frac=rf["hbm"]["frac"])
    # the whole step against the HBM roofline: algorithmic bytes of every stage / ms_per_step
    step_bytes = sum(layer_bytes(k, prec, act16, stems, masks16) for k in avg if k in LAYER_FLOP) * inst + (48.8 + 32.8 + stems * mask_pcm_kb) * 1024.0 * rows
    rf["step"]["hbm"] = {"achieved": step_bytes / (step_ms * 1e-3) / 1e9, "peak": PEAK_HBM_TBS * 1e3, "unit": "GB/s", "frac": step_bytes / (step_ms * 1e-3) / (PEAK_HBM_TBS * 1e12),
                         "algorithmic_bytes": step_bytes}
    for where, v in (("roofline", rf["frac"]), ("roofline.hbm", rf["hbm"]["frac"]), ("step", rf["step"]["frac"]), ("step.hbm", rf["step"]["hbm"]["frac"]), ("nn_stack", res["nn_stack"]["frac"])):
        assert nocheck or 0.0 < v <= 1.0, "roofline fraction %s = %g outside (0, 1]: wrong peak or wrong work count" % (where, v)
    assert nocheck or all(0.0 < v <= 1.0 for v in res["layer_executed_frac"].values()), res["layer_executed_frac"]
    if world == 1 and not a.no_cpu_baseline:
        try:
            res["cpu_baseline"] = cpu_baseline(a.cpu_tiles or None)
        except Exception as ex:                          # the baseline must never take the GPU number down with it
            res["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 0, "kind": "reference", "sample": "failed: %r" % (ex,)}
    return res


def flush_c_stdio():
    """librccl prints its version banner through C stdio, which is block-buffered when stdout is a pipe or a file: unflushed, it would land AFTER the JSON line at
    process exit.  The driver reads one JSON line from stdout - it has to be the last thing written."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launched():
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def visible_devices():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def ensure_world(a, argv):
    """--gpus N is binding.  Under a launcher: WORLD_SIZE must equal N.  Without one and N > 1 (per-process host): start the N ranks here with
    the same torch.distributed.run command the driver uses and exit with its status.  Never returns when it re-launches."""
    if launched():
        w = int(os.environ["WORLD_SIZE"])
        if w != a.gpus:
            raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (a.gpus, w))
        return
    if a.gpus == 1 or a.host == "native":
        return
    if not a.rendezvous_only:
        have = visible_devices()
        if have < a.gpus:
            raise SystemExit("bench.py: --gpus %d but %d GPU(s) visible: refusing to print a line for fewer devices than asked "
                             "(the HIP library has no CPU path)" % (a.gpus, have))
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def rendezvous_only(a):
    """CPU-testable half of the launcher path: join the world (gloo without a GPU), one all-reduce, rank 0 prints what it saw."""
    import torch
    import torch.distributed as dist
    from spleeterrt_amd import stream as srt_stream
    rank, world, on = srt_stream.init_distributed(None)
    t = torch.tensor([float(rank + 1)])
    if on:
        dist.all_reduce(t)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"rendezvous": True, "world": world, "n_gpus": a.gpus, "backend": dist.get_backend() if on else None, "rank_sum": float(t.item())}))
    if on:
        dist.destroy_process_group()


def measure_ranks(a, stems):
    """One process per GPU (this process is one rank).  Returns the measurement record make_line() prints, or None on ranks > 0."""
    import torch
    import torch.distributed as dist
    import spleeterrt_amd as srt

    from spleeterrt_amd import stream as srt_stream
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP library has no CPU path")
    if local >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no device (%d visible)" % (local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # under torch.distributed.run the process group is joined at ANY world size (nccl == RCCL on ROCm): a world of one still loads
    # RCCL and runs the broadcast / barrier / all-reduce below, so the N > 1 code path is exercised wherever the bench runs
    rank, world, dist_on = srt_stream.init_distributed(dev)
    assert world == a.gpus, (world, a.gpus)

    eng = srt.Engine(F=F, T=T, stem_modes=tuple(int(c) for c in a.stem_modes), oob_weights=(0.25, 0.0, 0.25, 0.25, 0.25)[:stems], variant=srt.VARIANT_VST,
                     max_tiles=a.tiles, impl=srt.IMPL_NAIVE if a.impl == "naive" else srt.IMPL_MFMA, device=dev,
                     precision={"f32": srt.PREC_F32, "f16": srt.PREC_F16, "f16x2": srt.PREC_F16X2}[a.precision])
    # weights: rank 0 creates them, one RCCL broadcast per blob (the only collective on this path)
    for s in range(stems):
        w = synth_weights(s, dev) if rank == 0 else torch.empty(9822725, device=dev)
        if dist_on:
            dist.broadcast(w, 0)
        eng.set_coeff(s, w)
    flush_c_stdio()                                          # (RCCL's banner, printed by the first collective, goes out now - not after the JSON line)
    n = a.tiles * T * HOP
    g = torch.Generator(device=dev).manual_seed(777 + rank)
    L = (torch.rand(n, device=dev, generator=g) - 0.5) * 0.2
    R = (torch.rand(n, device=dev, generator=g) - 0.5) * 0.2
    rows = eng.L.srtStftRows(n)
    out = torch.empty((stems, 2, eng.L.srtIstftLength(rows)), device=dev)

    def sync():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        eng.separate(L, R, out)
    sync()
    # the timed region: exactly K steps, nothing but the path's own launches on the stream (no per-launch events)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eng.separate(L, R, out)
    sync()
    dt = time.perf_counter() - t0
    # per-kernel durations: a second, separately timed pass of the same K steps with HIP events around every launch on the
    # engine's own stream (what `roofline` and `kernel_ms` are computed from; its wall time is reported as ms_per_step_events)
    eng.set_timing(True)
    t1 = time.perf_counter()
    for _ in range(a.steps):
        eng.separate(L, R, out)
    sync()
    dt_ev = time.perf_counter() - t1
    tim = eng.get_timing()
    kern = eng.get_timing_kernels()                           # [(launch name, kernel symbol the engine actually launched)]
    eng.set_timing(False)
    if dist_on:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    nocheck = os.environ.get("SRT_BENCH_NOCHECK") == "1"       # timing-ablation builds (scripts/gpu_tune.sh): wrong results by construction
    assert nocheck or torch.isfinite(out).all()
    rec = None
    if rank == 0:
        rec = {"dt": dt, "dt_ev": dt_ev, "tim": tim, "kern": kern, "world": world, "rows": rows, "frames_step": int(eng.L.srtStftFrames(n)),
               "host": "one process per GPU (torch.distributed)",
               "distributed": ({"backend": dist.get_backend() + " (RCCL)", "world": dist.get_world_size(),
                                "collectives": "%d weight-blob broadcasts at start-up; barrier + max-all-reduce around the timed region" % stems}
                               if dist_on else None)}
    eng.close()
    if dist_on:
        dist.destroy_process_group()
    return rec


def measure_native(a, stems):
    """ONE process: the C host (csrc/srt_multi.hip) owns N engines on N devices, one worker thread each - the shape of the reference's own
    fan-out (main.c:544-673).  Weights: one host blob per stem, uploaded once, ncclBroadcast over the devices (ncclCommInitAll)."""
    import ctypes as C
    import spleeterrt_amd as srt
    from spleeterrt_amd import capi
    have = visible_devices()
    if have < a.gpus:
        raise SystemExit("bench.py: --gpus %d --host native but %d GPU(s) visible" % (a.gpus, have))
    lib = srt.load_library()
    cfg = capi._Config()
    cfg.F, cfg.T, cfg.n_stems, cfg.variant, cfg.max_tiles = F, T, stems, srt.VARIANT_VST, a.tiles
    cfg.impl = srt.IMPL_NAIVE if a.impl == "naive" else srt.IMPL_MFMA
    cfg.precision = {"f32": srt.PREC_F32, "f16": srt.PREC_F16, "f16x2": srt.PREC_F16X2}[a.precision]
    for i in range(stems):
        cfg.stem_mode[i] = int(a.stem_modes[i])
        cfg.oob_weight[i] = (0.25, 0.0, 0.25, 0.25, 0.25)[i]
    m = C.c_void_p()

    def chk(rc):
        if rc < 0:
            raise SystemExit("libspleeterrt_amd: %s (rc=%d)" % (lib.srtLastError().decode(), rc))
        return rc
    chk(lib.srtMultiCreate(C.byref(cfg), None, a.gpus, C.byref(m)))
    for s in range(stems):
        w = synth_weights(s, "cpu").numpy()
        chk(lib.srtMultiSetCoeffHost(m, s, C.c_void_p(w.ctypes.data)))
    info = C.create_string_buffer(256)
    chk(lib.srtMultiInfo(m, info, len(info)))
    dt, dt_ev = C.c_double(), C.c_double()
    chk(lib.srtMultiBenchResident(m, a.tiles, a.steps, a.warmup, C.byref(dt), C.byref(dt_ev)))
    e0 = capi.Engine.__new__(capi.Engine)                     # borrowed view of engine 0 for the timing read-out (never closed)
    e0.L, e0.h = lib, C.c_void_p(lib.srtMultiEngine(m, 0))
    tim = e0.get_timing()
    kern = e0.get_timing_kernels()
    e0.set_timing(False)                                       # (srtMultiBenchResident leaves engine 0's timing window open for this read-out; close it)
    e0.h = None
    n = a.tiles * T * HOP
    text = info.value.decode()
    rec = {"dt": dt.value, "dt_ev": dt_ev.value, "tim": tim, "kern": kern, "world": a.gpus, "rows": lib.srtStftRows(n), "frames_step": int(lib.srtStftFrames(n)),
           "host": "native: one process, srtMultiCreate over %d device(s), one worker thread per engine (%s)" % (a.gpus, text),
           "distributed": {"backend": "rccl (ncclCommInitAll)" if "weights=rccl" in text else "hipMemcpyPeer (librccl unavailable)", "world": a.gpus,
                           "collectives": "%d weight-blob broadcasts at start-up; worker-thread barrier around the timed region" % stems}}
    lib.srtMultiDestroy(m)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tiles", type=int, default=TILES)
    ap.add_argument("--impl", default="mfma")
    ap.add_argument("--precision", default="f32", choices=["f32", "f16", "f16x2"],
                    help="conv contraction arithmetic; the headline metric is f32 (other modes are separate, labelled configurations)")
    ap.add_argument("--stems", type=int, default=STEMS, choices=[4, 5],
                    help="4 (default, the headline): BASELINE configs[2].  5 with --precision f16: BASELINE configs[4] (adds the piano sub-network)")
    ap.add_argument("--host", default="ranks", choices=["ranks", "native"],
                    help="ranks (default): one process per GPU over torch.distributed (bench.py starts them itself when no launcher did).  "
                         "native: one process, the C multi-device host (srtMulti*), one worker thread per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stem-modes", default=None, help="per stem: 1 = ELU/ELU (the 4-stem model), 0 = LeakyReLU/ReLU; measurement aid")
    ap.add_argument("--cpu-tiles", type=int, default=0)
    ap.add_argument("--rendezvous-only", action="store_true", help=argparse.SUPPRESS)     # tests: launcher + world join, no GPU work
    ap.add_argument("--config", default="c3", choices=["c3", "c4"],
                    help="c3 (default, the headline): BASELINE configs[2], 64-tile batches resident in HBM.  c4: BASELINE configs[3], the "
                         "60-minute host-resident stream partitioned by tile range over the ranks (scripts/stream_c4.py; PCIe-inclusive)")
    a = ap.parse_args()
    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    a.stem_modes = a.stem_modes or "1" * a.stems
    if len(a.stem_modes) != a.stems:
        raise SystemExit("bench.py: --stem-modes needs %d digits" % a.stems)
    ensure_world(a, sys.argv[1:])
    if a.rendezvous_only:
        return rendezvous_only(a)
    if a.config == "c4":
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import stream_c4
        reps = max(1, a.steps // 10)
        res, _ = stream_c4.run(max_tiles=a.tiles, gather=False, precision=a.precision, repeats=reps)
        if res is not None:                                   # the same line format as the headline; a step = one pass over the 60-minute stream
            line = {"metric": "x_realtime, PCIe-INCLUSIVE (4-stem separation of a 60-min 44.1 kHz stereo stream, host PCM -> host stems, tile-range partition)",
                    "value": res["x_realtime_pcie_inclusive"], "unit": "x real-time", "frames_per_s": res["frames_per_s"],
                    "n_gpus": res["n_gpus"], "steps": reps, "warmup": 1, "ms_per_step": res["seconds"] * 1e3, "higher_is_better": True,
                    "scaling": "strong", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
                    "config": {"workload": res["config"], "tiles_per_rank": res["tiles_per_rank"], "max_tiles_per_chunk": res["max_tiles_per_chunk"],
                               "parallelism": "tile-range partition x%d, weight broadcast only" % res["n_gpus"]},
                    "c4": res}
            flush_c_stdio()
            print(json.dumps(line), flush=True)
        return

    rec = measure_native(a, a.stems) if a.host == "native" else measure_ranks(a, a.stems)
    flush_c_stdio()
    if rec is not None:
        print(json.dumps(make_line(a, rec)), flush=True)


if __name__ == "__main__":
    main()
