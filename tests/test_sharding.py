"""CPU tests of the multi-GPU path (one process per GPU, tile-range sharding, single weight broadcast) with
world_size 2 on the gloo backend.  The engine is replaced by a CPU stand-in built on the oracle, so what is verified
is the host logic that the N>1 bench path relies on: plan() covers every tile exactly once, chunk + halo +
stitch reproduces the unsharded result, and the weight broadcast delivers identical blobs."""
import os
import socket

import numpy as np
import pytest

from spleeterrt_amd import stream


def test_plan_covers_stream_exactly_once():
    T = 64
    for n in (4096 * 9 + 8192, 4096 * 40 + 8192 + 5, 158769152):                 # last: the 60-minute C4 stream
        rows = stream.stft_rows(n)
        ntiles = (rows + T - 1) // T
        for world in (1, 2, 8):
            seen = []
            for r in range(world):
                for c in stream.plan(n, T, 7, r, world):
                    seen += list(range(c.tile0, c.tile1))
                    assert c.tile1 - c.tile0 <= 7 and c.rows <= (c.tile1 - c.tile0) * T and 0 <= c.frames <= c.rows
                    assert c.sample0 + c.nsamples <= n
            assert seen == list(range(ntiles))
    assert stream.stft_rows(158769152) == 155048 and (155048 + 255) // 256 == 606  # 606 tiles


class OracleEngine:
    """CPU stand-in with the Engine.separate_ex signature (tests only)."""

    def __init__(self, oracle, coeffs, modes, T, F, max_tiles):
        self.o, self.coeffs, self.modes, self.T, self.F, self.max_tiles = oracle, coeffs, modes, T, F, max_tiles

    def separate_ex(self, L, R, frames, rows):
        o = self.o
        L = np.ascontiguousarray(L); R = np.ascontiguousarray(R)
        n_fake = (frames - 1) * 1024 + 4096                                         # exactly `frames` transforms
        Lp = np.zeros(n_fake, np.float32); Rp = np.zeros(n_fake, np.float32)
        Lp[:min(L.size, n_fake)] = L[:n_fake]; Rp[:min(R.size, n_fake)] = R[:n_fake]
        re_f, im_f = o.stft(Lp, Rp)
        re = np.zeros((2, rows, 4096), np.float32); im = np.zeros_like(re)
        re[:, :frames] = re_f[:, :frames]; im[:, :frames] = im_f[:, :frames]
        outs = []
        for s, mode in enumerate(self.modes):
            r, i = re.copy(), im.copy()
            o.process_spectrogram(self.coeffs[s], r, i, self.F, self.T, mode, o.VARIANT_VST, 0.1)
            outs.append(o.istft(r, i))
        return np.stack(outs)


    def separate_host_stream(self, L, R, frames=None, rows=None, out=None, pinned=False):
        """Engine.separate_host_stream stand-in (what stream.separate_host_range / scripts/stream_c4.py call per rank)."""
        return self.separate_ex(L, R, frames, rows)


def test_rank_span_matches_plan():
    """rank_span (one span per rank, chunked natively by srtSeparateHostStream) covers exactly the chunks plan() lists."""
    T = 256
    n = 158769152
    for world in (1, 2, 8):
        offs = []
        for r in range(world):
            sp = stream.rank_span(n, T, r, world)
            cs = stream.plan(n, T, 64, r, world)
            assert sp.tile0 == cs[0].tile0 and sp.tile1 == cs[-1].tile1
            assert sp.sample0 == cs[0].sample0 and sp.rows == sum(c.rows for c in cs) and sp.frames == sum(c.frames for c in cs)
            assert sp.sample0 + sp.nsamples == cs[-1].sample0 + cs[-1].nsamples
            offs.append((sp.out_offset, sp.rows))
        for (o0, r0), (o1, _) in zip(offs, offs[1:]):
            assert o0 + r0 * 1024 == o1                                              # consecutive spans meet (+3072 overlap added when stitched)
        assert offs[-1][0] + offs[-1][1] * 1024 + 3072 == stream.total_output_length(n)


def test_c_partition_equals_rank_span():
    """srtRankSpan - the C partition of the native multi-device host (csrc/srt_multi.hip, what host/spleeterrt_cli.c fans out with) - is the
    same arithmetic as stream.rank_span for worlds 1 / 2 / 3 / 8 on the configs[3] length, a ragged short file, more ranks than tiles, and
    the minimum length.  Pure host code: runs without a device."""
    import ctypes as C
    import spleeterrt_amd
    from spleeterrt_amd.capi import Span as CSpan
    lib = spleeterrt_amd.load_library()
    lib.srtRankSpan.argtypes = [C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(CSpan)]
    for n, T in ((158769152, 256), (4096 * 108 + 8192, 64), (4096 * 108 + 8192, 256), (4096, 64), (1024 * 700 + 5, 128)):
        for world in (1, 2, 3, 8, 11):
            covered = 0
            for r in range(world):
                c = CSpan()
                assert lib.srtRankSpan(n, T, r, world, C.byref(c)) == 0
                sp = stream.rank_span(n, T, r, world)
                assert (c.tile0, c.tile1, c.sample0, c.nsamples, c.frames, c.rows, c.out_offset) == tuple(sp), (n, T, world, r)
                covered += c.rows
            assert covered == stream.stft_rows(n)
    c = CSpan()
    assert lib.srtRankSpan(4096, 64, 2, 2, C.byref(c)) < 0 and lib.srtRankSpan(4096, 64, 0, 0, C.byref(c)) < 0      # rank outside the world


def test_multi_device_host_refuses_without_a_gpu():
    """The native multi-device host has no CPU path either: without a HIP device srtMultiCreate fails with a code and a message (CPU suite only)."""
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import spleeterrt_amd
    from spleeterrt_amd.capi import _Config
    lib = spleeterrt_amd.load_library()
    cfg = _Config()
    cfg.F, cfg.T, cfg.n_stems, cfg.max_tiles = 512, 64, 2, 2
    h = C.c_void_p()
    assert lib.srtDeviceCount() == 0
    assert lib.srtMultiCreate(C.byref(cfg), None, 2, C.byref(h)) < 0 and b"no HIP device" in lib.srtLastError()
    assert lib.srtMultiCreate(C.byref(cfg), None, 0, C.byref(h)) < 0                     # bad worker count


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, T, F, q):
    import torch
    import torch.distributed as dist
    from oracle import pyoracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    coeffs = stream.broadcast_weights([O.synth_coeff(0)] if rank == 0 else None)
    c0 = coeffs[0].numpy()
    L, R = O.synth_audio(n, 777, True)
    eng = OracleEngine(O, [c0], (1,), T, F, max_tiles=1)
    parts = stream.separate_stream(eng, L, R, rank, world)
    sp, whole = stream.separate_host_range(eng, L, R, rank, world)                 # the configs[3] driver's per-rank call
    parts_range = [(sp.out_offset, whole)]
    gathered = [None] * world
    dist.all_gather_object(gathered, (float(np.abs(c0).sum()), parts, parts_range))
    if rank == 0:
        q.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_unsharded(oracle, coeffs):
    import torch.multiprocessing as mp
    T, F = 64, 512
    n = 4096 * 40 + 8192 + 300                                                       # 169 rows -> 3 tiles, ragged tail
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, T, F, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sums = [g[0] for g in gathered]
    assert sums[0] == sums[1] == float(np.abs(coeffs(0)).sum())                      # broadcast delivered the same blob
    parts = [pt for g in gathered for pt in g[1]]
    got = stream.stitch(parts, n, 1)
    L, R = oracle.synth_audio(n, 777, True)
    re, im = oracle.stft(L, R)
    oracle.process_spectrogram(coeffs(0), re, im, F, T, 1, oracle.VARIANT_VST, 0.1)
    ref = oracle.istft(re, im)
    assert got.shape[2] == ref.shape[1]
    assert np.abs(got[0] - ref).max() <= 2e-6 * np.abs(ref).max()                    # only the overlap-add order differs
    got2 = stream.stitch([pt for g in gathered for pt in g[2]], n, 1)                # one span per rank (scripts/stream_c4.py)
    assert np.abs(got2[0] - ref).max() <= 2e-6 * np.abs(ref).max()
