#!/usr/bin/env python3
"""BASELINE configs[3]: a 60-minute synthetic stereo stream, 4 stems, partitioned by tile range over the ranks.

One process per GPU.  Per rank: weights arrive by ONE broadcast per blob (RCCL when world > 1; the only collective),
`stream.rank_span(rank, world)` gives its contiguous tile range (+ 3072-sample halo), and one srtSeparateHostStream
call runs it: chunks of --max-tiles tiles, H2D / compute / D2H overlapped on three HIP streams, chunk overlaps carried on
the device.  Rank 0 then collects the per-rank parts and adds the 3072-sample seams (stream.stitch semantics).  This is
the reference's processMT fan-out (Executable/main.c:544-673: tile ranges -> workers -> join) with GPUs as the workers.

    python scripts/stream_c4.py                                   # 1 GPU, all 606 tiles
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 scripts/stream_c4.py

Host buffers are page-locked (torch pin_memory), so the timed region is upload + compute + download of every sample: the
reported x real-time is PCIe-INCLUSIVE (bench.py's headline `value` is HBM-resident and never this number).
Prints one JSON line; --out writes it to a file (profiles/r02_c4.json).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

F, T, STEMS = 1024, 256, 4
FS, HOP = 44100.0, 1024
BLOCK = 1 << 22


def synth_stream(n, lo=0, hi=None, out=None):
    """Seeded stereo noise +-0.1 plus three tones (SURVEY §8d's signal class), a pure function of the sample index:
    block b of 4 Mi samples comes from its own Philox key, so any rank can generate any span of the same stream."""
    hi = n if hi is None else hi
    L = np.empty(hi - lo, np.float32) if out is None else out[0]
    R = np.empty(hi - lo, np.float32) if out is None else out[1]
    b0, b1 = lo // BLOCK, (hi + BLOCK - 1) // BLOCK
    # 220, 1760 and 7040 Hz at 44.1 kHz share the period 2205 samples: one table, indexed modulo
    k = np.arange(2205, dtype=np.float64) / FS
    period = (0.05 * (np.sin(2 * np.pi * 220.0 * k) + np.sin(2 * np.pi * 1760.0 * k) + np.sin(2 * np.pi * 7040.0 * k))).astype(np.float32)
    for b in range(b0, b1):
        g = np.random.Generator(np.random.Philox(key=[777, b]))
        blk = (g.random((2, BLOCK), dtype=np.float32) - 0.5) * 0.2
        s, e = max(lo, b * BLOCK), min(hi, (b + 1) * BLOCK)
        tone = np.resize(np.roll(period, -(s % 2205)), e - s)
        L[s - lo:e - lo] = blk[0, s - b * BLOCK:e - b * BLOCK] + tone
        R[s - lo:e - lo] = blk[1, s - b * BLOCK:e - b * BLOCK] + tone
    return L, R


def run(minutes=60.0, max_tiles=64, gather=True, check_seams=True, precision="f32", repeats=1):
    import torch
    import torch.distributed as dist
    import spleeterrt_amd as srt
    from spleeterrt_amd import stream
    from bench import synth_weights

    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("stream_c4.py needs a GPU: the HIP library has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rank, world, dist_on = stream.init_distributed(dev)          # joins the nccl (RCCL) group under a launcher at any world size, 1 included

    n_audio = int(round(minutes * 60 * FS))
    n = 4096 * ((n_audio + 4095) // 4096) + 8192                  # the CLI's padding (main.c:762-767): 60 min -> 158 769 152
    rows = stream.stft_rows(n)
    ntiles = (rows + T - 1) // T

    eng = srt.Engine(F=F, T=T, stem_modes=(1,) * STEMS, oob_weights=(0.25, 0.0, 0.25, 0.25), variant=srt.VARIANT_VST,
                     max_tiles=max_tiles, device=dev,
                     precision={"f32": srt.PREC_F32, "f16": srt.PREC_F16, "f16x2": srt.PREC_F16X2}[precision])
    t_b0 = time.perf_counter()
    for s in range(STEMS):                                          # the one collective of the path
        w = synth_weights(s, dev) if rank == 0 else torch.empty(9822725, device=dev)
        if dist_on:
            dist.broadcast(w, 0)
        eng.set_coeff(s, w)
    torch.cuda.synchronize()
    t_bcast = time.perf_counter() - t_b0                          # synthesis on rank 0 + broadcast + GEMM packing (outside the timed region)

    # this rank's span of the stream, generated in place into page-locked memory (a rank never touches the rest)
    sp = stream.rank_span(n, T, rank, world)
    Lp = torch.empty(max(sp.nsamples, 1), dtype=torch.float32, pin_memory=True)
    Rp = torch.empty(max(sp.nsamples, 1), dtype=torch.float32, pin_memory=True)
    Lv, Rv = Lp.numpy(), Rp.numpy()
    body = min(n_audio, sp.sample0 + sp.nsamples) - sp.sample0       # the padding tail is silence
    Lv[:] = 0.0
    Rv[:] = 0.0
    if body > 0:
        synth_stream(n_audio, sp.sample0, sp.sample0 + body, out=(Lv[:body], Rv[:body]))
    out_len = sp.rows * HOP + 3072
    outp = torch.empty(STEMS * 2 * max(out_len, 1), dtype=torch.float32, pin_memory=True)

    def sync():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()

    def one_pass():
        if sp.rows:
            eng.separate_host_stream(Lp[:sp.nsamples], Rp[:sp.nsamples], frames=sp.frames, rows=sp.rows, out=outp, pinned=True)

    one_pass()                                                       # warm-up: staging buffers, kernels, page tables
    sync()
    t0 = time.perf_counter()
    for _ in range(repeats):
        one_pass()
    sync()
    dt = (time.perf_counter() - t0) / repeats
    if dist_on:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    part = outp.numpy()[:STEMS * 2 * out_len].reshape(STEMS, 2, out_len)
    res = None
    # collection on rank 0 (outside the timed region: a deployment would write per-rank files or stream them on)
    t_g0 = time.perf_counter()
    full = None
    if gather:
        total = stream.total_output_length(n)
        if rank == 0:
            full = np.zeros((STEMS, 2, total), np.float32)
            full[:, :, sp.out_offset:sp.out_offset + out_len] += part
            for r in range(1, world):
                spr = stream.rank_span(n, T, r, world)
                if spr.rows == 0:
                    continue
                lr = spr.rows * HOP + 3072
                buf = torch.empty(STEMS * 2 * lr, device=dev)
                dist.recv(buf, src=r)
                full[:, :, spr.out_offset:spr.out_offset + lr] += buf.cpu().numpy().reshape(STEMS, 2, lr)
                del buf
        elif sp.rows:
            dist.send(outp[:STEMS * 2 * out_len].to(dev), dst=0)
    t_gather = time.perf_counter() - t_g0

    if rank == 0:
        frames = rows
        res = {
            "config": "BASELINE configs[3]: 4-stem, %.1f-min synthetic stereo stream (%d samples padded, %d rows, %d tiles of %dx%d), tile-range partition over %d GPU(s)"
                      % (minutes, n, rows, ntiles, T, F, world),
            "n_gpus": world, "tiles": ntiles, "rows": rows, "max_tiles_per_chunk": max_tiles, "precision": precision,
            "tiles_per_rank": [stream.rank_span(n, T, r, world).tile1 - stream.rank_span(n, T, r, world).tile0 for r in range(world)],
            "seconds": dt, "frames_per_s": frames / dt, "x_realtime_pcie_inclusive": frames * HOP / FS / dt,
            "timed_region": "page-locked host PCM -> H2D -> STFT/U-Nets/mask/iSTFT -> D2H page-locked host stems, max over ranks, %d repeat(s) after one warm-up pass" % repeats,
            "bytes_h2d": 2 * 4 * n, "bytes_d2h": STEMS * 2 * 4 * (rows * HOP + 3072),
            "weight_setup_s": t_bcast, "collect_on_rank0_s": t_gather if gather else None,
            "collective": "broadcast of %d weight blobs (39.29 MB each) only" % STEMS,
            "process_group": (dist.get_backend() if dist_on else None),
        }
        if full is not None:
            res["checksum"] = {"sum": [float(full[s].astype(np.float64).sum()) for s in range(STEMS)],
                               "sumsq": [float((full[s].astype(np.float64) ** 2).sum()) for s in range(STEMS)],
                               "peak": float(np.abs(full).max()), "finite": bool(np.isfinite(full).all())}
    eng.close()
    return res, full


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=60.0)
    ap.add_argument("--max-tiles", type=int, default=64)
    ap.add_argument("--precision", default="f32", choices=["f32", "f16", "f16x2"])
    ap.add_argument("--repeats", type=int, default=1)
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    res, _ = run(a.minutes, a.max_tiles, gather=not a.no_gather, precision=a.precision, repeats=a.repeats)
    if res is not None:
        line = json.dumps(res)
        print(line)
        if a.out:
            os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
            open(a.out, "w").write(json.dumps(res, indent=1) + "\n")
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
