"""BASELINE configs[3] at full size on one GPU: the 60-minute synthetic stereo stream (158 769 152 padded samples,
155 048 rows, 606 tiles of 256x1024, 4 stems) through srtSeparateHostStream — the per-rank worker of the tile-range
partition (scripts/stream_c4.py; the reference's processMT fan-out, Executable/main.c:544-673).

What is checked (sizes the CPU oracle cannot follow are covered by size-independent properties):
  (a) chunk-size invariance: 64-tile chunks == 37-tile chunks (606 = 16*37 + 14: ragged last chunk) to 2e-6 * peak —
      tiles are independent, so only the order of the overlap-add at chunk seams may differ;
  (b) three sampled tiles — the first, the first tile after a chunk boundary, the ragged last one — against the CPU
      oracle's stft -> processMT -> istft of just their own PCM span (stems rel-RMS <= 1e-4);
  (c) the world=2 plan (two rank spans run one after the other on this GPU, parts added at the 3072-sample seam)
      == world=1 to 2e-6 * peak: the N-rank path differs from N=1 only by where the stream is cut.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

T, F, S, HOP = 256, 1024, 4, 1024
MODES = (1, 1, 1, 1)
OOB = (0.25, 0.0, 0.25, 0.25)


def _rel_rms(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2)) / (np.sqrt(np.mean(b ** 2)) + 1e-30))


def test_c4_sixty_minute_stream_one_gpu(oracle, coeffs):
    import torch
    import spleeterrt_amd as srt
    from spleeterrt_amd import stream
    import stream_c4

    n_audio = 60 * 60 * 44100
    n = 4096 * ((n_audio + 4095) // 4096) + 8192
    assert n == 158769152
    rows = stream.stft_rows(n)
    ntiles = (rows + T - 1) // T
    assert rows == 155048 and ntiles == 606
    total = stream.total_output_length(n)

    Lp = torch.zeros(n, dtype=torch.float32, pin_memory=True)
    Rp = torch.zeros(n, dtype=torch.float32, pin_memory=True)
    L, R = Lp.numpy(), Rp.numpy()
    stream_c4.synth_stream(n_audio, 0, n_audio, out=(L[:n_audio], R[:n_audio]))

    def engine(max_tiles):
        e = srt.Engine(F=F, T=T, stem_modes=MODES, oob_weights=OOB, variant=srt.VARIANT_VST, max_tiles=max_tiles)
        for s in range(S):
            e.set_coeff(s, coeffs(s))
        return e

    # world = 1, chunks of 64 tiles
    e64 = engine(64)
    outA = torch.empty(S * 2 * total, dtype=torch.float32, pin_memory=True)
    e64.separate_host_stream(Lp, Rp, out=outA, pinned=True)
    A = outA.numpy().reshape(S, 2, total)
    assert np.isfinite(A[:, :, ::997]).all()
    peak = max(float(np.abs(A[s, c]).max()) for s in range(S) for c in range(2))
    assert 0.05 < peak < 10.0

    # (b) sampled tiles against the CPU oracle: interior samples [j*T*1024 + 3072, (j+1)*T*1024) depend on tile j's frames only
    for j, stems in ((0, range(S)), (64, (1, 3)), (ntiles - 1, (0, 2))):
        s0 = j * T * HOP
        s1 = min(n, (j + 1) * T * HOP + 3072)
        re, im = oracle.stft(np.ascontiguousarray(L[s0:s1]), np.ascontiguousarray(R[s0:s1]))
        if j < ntiles - 1:
            re, im = re[:, :T].copy(), im[:, :T].copy()        # the 3 halo rows belong to the next tile
        lo, hi = 3072, re.shape[1] * HOP
        if j == 0:
            lo = 0                                              # nothing precedes the first tile
        if j == ntiles - 1:
            hi = re.shape[1] * HOP + 3072                       # nothing follows the last one
        for s in stems:
            r, i = re.copy(), im.copy()
            oracle.process_spectrogram(coeffs(s), r, i, F, T, MODES[s], oracle.VARIANT_VST, OOB[s])
            ref = oracle.istft(r, i)[:, lo:hi]
            got = A[s][:, s0 + lo:s0 + hi]
            err = _rel_rms(got, ref)
            assert err <= 1e-4, "tile %d stem %d: rel rms %g" % (j, s, err)
            assert np.abs(got - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-6)

    # (a) chunk-size invariance
    e37 = engine(37)
    outB = torch.empty(S * 2 * total, dtype=torch.float32, pin_memory=True)
    e37.separate_host_stream(Lp, Rp, out=outB, pinned=True)
    B = outB.numpy().reshape(S, 2, total)
    for s in range(S):
        for c in range(2):
            d = float(np.abs(A[s, c] - B[s, c]).max())
            assert d <= 2e-6 * peak, "chunk-size invariance stem %d ch %d: %g (peak %g)" % (s, c, d, peak)
    e37.close()
    del B, outB

    # (c) the two-rank plan, both spans run here, stitched at the seam
    sp0, p0 = stream.separate_host_range(e64, Lp, Rp, 0, 2, pinned=True)
    sp1, p1 = stream.separate_host_range(e64, Lp, Rp, 1, 2, pinned=True)
    assert sp0.tile0 == 0 and sp0.tile1 == sp1.tile0 == 303 and sp1.tile1 == 606
    off = sp1.out_offset
    assert p0.shape[2] == off + 3072 and off + p1.shape[2] == total
    tol = 2e-6 * peak
    for s in range(S):
        for c in range(2):
            assert float(np.abs(A[s, c, :off] - p0[s, c, :off]).max()) <= tol
            assert float(np.abs(A[s, c, off:off + 3072] - (p0[s, c, off:] + p1[s, c, :3072])).max()) <= tol
            assert float(np.abs(A[s, c, off + 3072:] - p1[s, c, 3072:]).max()) <= tol
    e64.close()


def test_c4_driver_script_small_stream():
    """scripts/stream_c4.py itself (the N-rank driver, here at world = 1) on a 30-second stream: its gathered output equals one
    device-resident srtSeparate of the same stream, and its report carries the metric fields."""
    import torch
    import spleeterrt_amd as srt
    import stream_c4
    from bench import synth_weights
    res, full = stream_c4.run(minutes=0.5, max_tiles=2, gather=True)
    assert res["n_gpus"] == 1 and res["tiles"] == sum(res["tiles_per_rank"]) and res["checksum"]["finite"]
    assert res["x_realtime_pcie_inclusive"] > 0 and res["bytes_d2h"] == full.size * 4
    n_audio = int(round(0.5 * 60 * 44100))
    n = 4096 * ((n_audio + 4095) // 4096) + 8192
    L, R = stream_c4.synth_stream(n_audio)
    Lp = np.zeros(n, np.float32); Rp = np.zeros(n, np.float32)
    Lp[:n_audio] = L; Rp[:n_audio] = R
    dev = torch.device("cuda", 0)
    eng = srt.Engine(F=F, T=T, stem_modes=MODES, oob_weights=OOB, variant=srt.VARIANT_VST, max_tiles=res["tiles"], device=dev)
    for s in range(S):
        eng.set_coeff(s, synth_weights(s, dev))
    ref = eng.separate(torch.from_numpy(Lp).cuda(), torch.from_numpy(Rp).cuda()).cpu().numpy()
    eng.close()
    assert ref.shape == full.shape
    assert np.abs(full - ref).max() <= 1e-5 * np.abs(ref).max()      # chunk seams + (2-tile chunks) split-K association
