#!/usr/bin/env python3
"""two engines on ONE device.  The native multi-device host (csrc/srt_multi.hip) with the device list (0, 0): two engines, two host
threads, two HIP streams on one GPU, 32 tiles each - against one engine with 64 tiles, same box, same process.  If one half-batch's bandwidth kernels and kernel
tails fill behind the other's MFMA kernels, the pair finishes 64 tiles sooner than the single engine does.

    python scripts/two_on_one.py [--precision f32] [--steps 20] [--out gpurun_out/x.json]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import spleeterrt_amd as srt
from spleeterrt_amd import capi


def run(lib, devices, tiles, stems, precision, steps, warmup):
    cfg = capi._Config()
    cfg.F, cfg.T, cfg.n_stems, cfg.variant, cfg.max_tiles = bench.F, bench.T, stems, srt.VARIANT_VST, tiles
    cfg.impl, cfg.precision = srt.IMPL_MFMA, {"f32": srt.PREC_F32, "f16": srt.PREC_F16, "f16x2": srt.PREC_F16X2}[precision]
    for i in range(stems):
        cfg.stem_mode[i] = 1
        cfg.oob_weight[i] = (0.25, 0.0, 0.25, 0.25, 0.25)[i]
    m = C.c_void_p()
    dev = (C.c_int * len(devices))(*devices)
    if lib.srtMultiCreate(C.byref(cfg), dev, len(devices), C.byref(m)) < 0:
        raise SystemExit(lib.srtLastError().decode())
    for s in range(stems):
        w = bench.synth_weights(s, "cpu").numpy()
        if lib.srtMultiSetCoeffHost(m, s, C.c_void_p(w.ctypes.data)) < 0:
            raise SystemExit(lib.srtLastError().decode())
    dt = C.c_double()
    if lib.srtMultiBenchResident(m, tiles, steps, warmup, C.byref(dt), None) < 0:
        raise SystemExit(lib.srtLastError().decode())
    lib.srtMultiDestroy(m)
    n = tiles * bench.T * bench.HOP
    frames = int(lib.srtStftFrames(n)) * len(devices) * steps
    return {"engines": len(devices), "tiles_per_engine": tiles, "ms_per_64_tiles": dt.value / steps * 1e3 * 64.0 / (tiles * len(devices)),
            "frames_per_s": frames / dt.value, "x_realtime": frames / dt.value * bench.HOP / bench.FS}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--stems", type=int, default=4)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    lib = srt.load_library()
    res = {"precision": a.precision, "stems": a.stems, "runs": []}
    for devices, tiles in (([0], 64), ([0, 0], 32), ([0], 64), ([0, 0], 32), ([0, 0, 0, 0], 16)):
        r = run(lib, devices, tiles, a.stems, a.precision, a.steps, a.warmup)
        res["runs"].append(r)
        print(r, flush=True)
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
