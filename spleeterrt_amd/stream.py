"""Tile-range partitioning of a long stereo stream: chunks for one GPU, shards for N GPUs.

The reference processes a spectrogram as independent, non-overlapping T-frame tiles with no cross-tile state
(Executable/main.c:455-495); its own multi-thread mode hands contiguous tile ranges to threads (main.c:545-575).
Here the same ranges go to GPUs (one process per GPU) and, inside a GPU, to batches of at most `max_tiles` tiles.
A range [j0, j1) needs the PCM samples [j0*T*1024, j1*T*1024 + 3072) and produces the overlap-add contribution of
its own frames, which covers output samples [j0*T*1024, j1*T*1024 + 3072): consecutive ranges overlap by 3072
samples and are simply ADDED when stitched.  No data-path collective exists; weights are broadcast once.
"""
from collections import namedtuple

HOP, FFT = 1024, 4096

Chunk = namedtuple("Chunk", "tile0 tile1 sample0 nsamples frames rows out_offset")


def stft_rows(n):
    return (n + HOP - 1) // HOP                                   # stftFix.c:367


def stft_frames(n):
    return 0 if n < FFT else (n - FFT + HOP // 4) // HOP + 1      # stftFix.c:378 + the tail frame


def rank_tiles(ntiles, rank, world):
    """Contiguous tile range of one rank (GPU g gets tiles [g*ceil(N/G), ...))."""
    per = (ntiles + world - 1) // world
    return min(rank * per, ntiles), min((rank + 1) * per, ntiles)


def plan(n, T, max_tiles, rank=0, world=1):
    """Chunks (<= max_tiles tiles each) this rank has to run for an n-sample stream."""
    rows, frames = stft_rows(n), stft_frames(n)
    ntiles = (rows + T - 1) // T
    t0, t1 = rank_tiles(ntiles, rank, world)
    out = []
    j = t0
    while j < t1:
        j1 = min(j + max_tiles, t1)
        row0, row1 = j * T, min(j1 * T, rows)
        s0 = row0 * HOP
        ns = min(n, row1 * HOP + (FFT - HOP)) - s0
        out.append(Chunk(j, j1, s0, ns, max(0, min(frames - row0, row1 - row0)), row1 - row0, s0))
        j = j1
    return out


Span = namedtuple("Span", "tile0 tile1 sample0 nsamples frames rows out_offset")


def rank_span(n, T, rank=0, world=1):
    """The contiguous tile range of one rank as ONE span (what srtSeparateHostStream chunks natively): tiles
    [tile0, tile1), PCM samples [sample0, sample0 + nsamples) (3072-sample halo included), `frames` transformed
    frames, `rows` rows, and the offset of its overlap-add contribution in the full output."""
    rows, frames = stft_rows(n), stft_frames(n)
    ntiles = (rows + T - 1) // T
    t0, t1 = rank_tiles(ntiles, rank, world)
    row0, row1 = t0 * T, min(t1 * T, rows)
    s0 = row0 * HOP
    ns = max(0, min(n, row1 * HOP + (FFT - HOP)) - s0)
    return Span(t0, t1, s0, ns, max(0, min(frames - row0, row1 - row0)), max(0, row1 - row0), s0)


def separate_host_range(engine, L, R, rank=0, world=1, out=None, pinned=False):
    """This rank's share of a HOST-resident stream (numpy / pinned arrays of the whole stream, or anything sliceable):
    one srtSeparateHostStream call over its tile range — chunks of engine.max_tiles tiles, H2D / compute / D2H overlapped
    on three HIP streams, chunk overlaps carried on the device.  Returns (span, stems [S,2,rows*1024+3072]) or
    (span, None) when the rank has no tiles.  The reference's counterpart is one tile-range worker of processMT
    (Executable/main.c:544-673); there is no exchange with other ranks."""
    sp = rank_span(len(L), engine.T, rank, world)
    if sp.rows == 0:
        return sp, None
    o = engine.separate_host_stream(L[sp.sample0:sp.sample0 + sp.nsamples], R[sp.sample0:sp.sample0 + sp.nsamples],
                                    frames=sp.frames, rows=sp.rows, out=out, pinned=pinned)
    return sp, o


def total_output_length(n):
    return stft_rows(n) * HOP + (FFT - HOP)                       # stftFix.c:500


def separate_stream(engine, L, R, rank=0, world=1):
    """Run this rank's chunks.  `engine` needs .T, .max_tiles and .separate_ex(L, R, frames, rows) -> [S,2,rows*1024+3072]
    (spleeterrt_amd.Engine on a GPU; tests substitute a CPU stand-in).  Returns [(out_offset, array)]."""
    parts = []
    for c in plan(len(L), engine.T, engine.max_tiles, rank, world):
        o = engine.separate_ex(L[c.sample0:c.sample0 + c.nsamples], R[c.sample0:c.sample0 + c.nsamples], c.frames, c.rows)
        parts.append((c.out_offset, o))
    return parts


def stitch(parts, n, nstems):
    """Add the (offset, [S,2,len]) contributions of all chunks / ranks into the full-length output (host side)."""
    import numpy as np
    out = np.zeros((nstems, 2, total_output_length(n)), np.float32)
    for off, o in sorted(parts, key=lambda p: p[0]):
        o = o.detach().cpu().numpy() if hasattr(o, "detach") else np.asarray(o)
        out[:, :, off:off + o.shape[2]] += o
    return out


def init_distributed(device=None):
    """One process per GPU.  Under a torch.distributed launcher (RANK / WORLD_SIZE / MASTER_ADDR in the environment) join the
    process group - `nccl` (= RCCL on ROCm) when `device` is a GPU, gloo otherwise - at ANY world size, 1 included: a world of one
    still loads RCCL and runs the weight broadcast, so the N > 1 code path is the one that always runs.  Returns (rank, world, on)."""
    import os
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ and "MASTER_ADDR" in os.environ
    if (world > 1 or launched) and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC
        if device is not None and str(device).startswith("cuda"):
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
    return rank, world, dist.is_initialized()


def broadcast_weights(coeffs, device=None, src=0):
    """The one collective of the path: rank `src` holds the blobs (list of float32 arrays/tensors, one per stem),
    every rank gets them (RCCL over xGMI when the process group is `nccl`, gloo on CPU)."""
    import torch
    import torch.distributed as dist
    n = 9822725
    k = [len(coeffs) if coeffs is not None else 0]
    if dist.is_initialized():
        dist.broadcast_object_list(k, src)
    out = []
    for s in range(k[0]):
        if coeffs is not None:
            t = torch.as_tensor(coeffs[s], dtype=torch.float32).reshape(-1).to(device or "cpu")
        else:
            t = torch.empty(n, dtype=torch.float32, device=device or "cpu")
        if dist.is_initialized():
            dist.broadcast(t, src)
        out.append(t)
    return out
