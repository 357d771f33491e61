"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

Tolerances (fp32 path, BASELINE.md §4 / SURVEY §8d): mask max-abs <= 2e-4 with the exact sigmoid
(<= 1e-3 for the LUT flavour, whose table has a 9e-4 step at |x| = 7), stems rel-RMS <= 1e-4 and
max-abs <= 1e-4 * peak.  Intermediate tensors are checked relative to their own RMS.
"""
import contextlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MASK_TOL_EXACT = 2e-4
MASK_TOL_LUT = 1e-3


def _engine(**kw):
    import spleeterrt_amd as srt
    return srt.Engine(**kw)


def _mag_input(oracle, ntiles, T, F, seed=4242):
    # magnitude-like, positive, with a few strong peaks (same order of magnitude as |STFT|*4096 of +-0.1 noise)
    x = np.abs(oracle.lcg(seed, ntiles * 2 * T * F, 6.0)).reshape(ntiles, 2, T, F)
    x[:, :, ::7, ::13] *= 8.0
    return np.ascontiguousarray(x, np.float32)


def _rel_rms(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2)) / (np.sqrt(np.mean(b ** 2)) + 1e-30))


TAP_RMS_TOL = 2e-5          # every intermediate tensor, relative to its own RMS
TAP_MAX_TOL = 1e-4          # ... and its worst single element relative to the tensor's peak: a handful of wrong border pixels in a
                            # multi-million-element tensor passes an RMS bound, it cannot pass this one


def _check_taps(eng, oracle, coeff, x_tile, mode, s, t, masks=None, tag=""):
    """every tensor of instance (stem s, tile t) against the oracle: rel-RMS AND max-abs / peak; returns the worst of each"""
    y, taps = oracle.forward(coeff, x_tile, mode, oracle.VARIANT_VST, want_taps=True)
    worst_r = worst_m = 0.0
    for name, ref in taps.items():
        got = eng.tensor(name, s, t)
        err = _rel_rms(got, ref)
        mx = float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))
        where = np.unravel_index(int(np.abs(got - ref).argmax()), ref.shape)
        assert err < TAP_RMS_TOL, "%s %s stem %d tile %d: rel rms %g (max abs / peak %g at %r)" % (tag, name, s, t, err, mx, where)
        assert mx < TAP_MAX_TOL, "%s %s stem %d tile %d: max abs / peak %g at %r of %r (rel rms %g)" % (tag, name, s, t, mx, where, ref.shape, err)
        worst_r, worst_m = max(worst_r, err), max(worst_m, mx)
    if masks is not None:
        d = float(np.abs(masks[s, t] - y).max())
        assert d <= MASK_TOL_EXACT, "%s mask stem %d tile %d: max abs %g" % (tag, s, t, d)
    return worst_r, worst_m


@contextlib.contextmanager
def _env(**kv):
    """environment switches the engine reads per forward, restored on exit"""
    import os
    old = {k: os.environ.get(k) for k in kv}
    try:
        for k, v in kv.items():
            os.environ[k] = str(v)
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _layer_kernels(eng, xd):
    """layer name -> kernel symbol the engine launched for it in one forward (srtGetTimingKernels)"""
    eng.set_timing(True)
    eng.forward(xd)
    ks = dict(eng.get_timing_kernels())
    eng.set_timing(False)
    return ks


@pytest.mark.parametrize("impl", ["naive", "mfma"])
@pytest.mark.parametrize("T,F", [(64, 512), (128, 1024)])
def test_forward_layers(oracle, coeffs, impl, T, F):
    import torch
    import spleeterrt_amd as srt
    modes = (0, 1)                       # stem 0: LeakyReLU/ReLU ("2stems" vocal), stem 1: ELU/ELU
    ntiles = 3 if T == 64 else 1
    eng = _engine(F=F, T=T, stem_modes=modes, variant=srt.VARIANT_VST, max_tiles=4,
                  impl=srt.IMPL_NAIVE if impl == "naive" else srt.IMPL_MFMA)
    for s in range(2):
        eng.set_coeff(s, coeffs(s))
    x = _mag_input(oracle, ntiles, T, F)
    masks = eng.forward(torch.from_numpy(x).cuda()).cpu().numpy()
    worst = (0.0, 0.0)
    for s in range(2):
        for t in range(ntiles):
            worst = max(worst, _check_taps(eng, oracle, coeffs(s), x[t], modes[s], s, t, masks, impl))
    eng.close()
    print("forward %s %dx%d worst tap rel-rms %.3g, max-abs/peak %.3g" % (impl, T, F, worst[0], worst[1]))


@pytest.mark.parametrize("T,F,check", [(64, 512, (0, 4, 8)), (128, 1024, (8,))])
def test_winograd_decoder_layers(oracle, coeffs, T, F, check):
    """Batches above 16 instances run the decoder layers up2..up5 in their Winograd form (csrc/srt_nn4.hip: F(2,3)/F(2,2) over 2x2 input
    blocks, 49 MFMA products per block instead of 100).  9 tiles x 2 stems = 18 instances: every tensor of the sampled tiles against the
    oracle at the same tolerance as the direct kernels.  9 is not a multiple of the 4-instance tile of the deep layers (a partly empty
    instance group), T = 64 leaves up2 below the smallest tile (it stays on the direct kernel), T = 128 covers up2."""
    import torch
    import spleeterrt_amd as srt
    modes = (0, 1)
    ntiles = 9
    eng = _engine(F=F, T=T, stem_modes=modes, variant=srt.VARIANT_VST, max_tiles=ntiles, impl=srt.IMPL_MFMA)
    for s in range(2):
        eng.set_coeff(s, coeffs(s))
    x = _mag_input(oracle, ntiles, T, F, seed=99)
    xd = torch.from_numpy(x).cuda()
    masks = eng.forward(xd).cpu().numpy()
    assert np.isfinite(masks).all()
    for s in range(2):
        for t in check:
            _check_taps(eng, oracle, coeffs(s), x[t], modes[s], s, t, masks)
    ks = _layer_kernels(eng, xd)
    for name in ("up3", "up4", "up5") + (("up2",) if T >= 128 else ()):
        assert ks[name].startswith("srt_dec_wino"), (name, ks[name])
    assert not ks["up1"].startswith("srt_dec_wino")
    eng.close()


def _wino_expected(T, F, lvl):
    """does up<lvl> (lvl = 1..5) of a T x F engine fit the Winograd kernels' geometry (csrc/srt_nn4.hip: H even, W % 4 == 0, H >= 4, W >= 16)"""
    H, W = T >> (7 - lvl), F >> (7 - lvl)
    return H % 2 == 0 and W % 4 == 0 and H >= 4 and W >= 16


def _enc_wino_expected(T, F, lvl):
    """does down<lvl> of a T x F engine run in Winograd form (csrc/srt_nn4.hip srt_enc_wino_covers: Cin >= 32, i.e. down3..down6; input H, W
    multiples of 4; output at least 4 x 16)"""
    H, W = T >> (lvl - 1), F >> (lvl - 1)
    return lvl >= 3 and H % 4 == 0 and W % 4 == 0 and H // 2 >= 4 and W // 2 >= 16


@pytest.mark.parametrize("T,F,ntiles,stems,check_stems", [
    (256, 1536, 5, 4, (0, 3)),      # the plugin's geometry (PluginProcessor.cpp:124) as a 20-instance batch: up1 4 x 24 (two instances per workgroup, 5 tiles: a half-empty pair), up2 8 x 48 (half-empty x tile)
    (192, 320, 9, 2, (0, 1)),       # up2 W = 10 (not a multiple of 4: direct kernel), up3 12 x 20 on the 4-instance tile with a partly empty group
    (64, 576, 9, 2, (0, 1)),        # up2 2 x 18 -> direct, up3 4 x 36 (smallest height), up4 8 x 72, up5 16 x 144: partial x tiles everywhere
    (320, 1984, 5, 4, (1, 2)),      # up2 10 x 62 -> direct; up3 20 x 124, up4 40 x 248, up5 80 x 496: partial tiles in both directions
    (128, 1024, 17, 1, (0,)),       # one stem, odd tile count just over the 16-instance switch: ragged instance groups, tpw > 1
])
def test_winograd_decoder_odd_geometries(oracle, coeffs, T, F, ntiles, stems, check_stems):
    """the Winograd kernels (decoder up1..up5, and since round 3 the encoder's down3..down6) on batches above 16 instances at NON-power-of-two
    geometries: partial spatial tiles (W % 32 != 0, H % 8 != 0: the out-of-range zero fill and the blk_ok store guard), the
    4-instance tile with a partly empty instance group, several units per workgroup, and the clean fall-back to the direct kernels
    where W % 4 != 0.  Every tensor of the first, the last and one interior tile: rel-RMS and max-abs / peak; the kernel that ran
    each decoder layer is read back from the engine and must be the Winograd one exactly where the geometry fits."""
    import torch
    import spleeterrt_amd as srt
    modes = tuple((s + 1) % 2 for s in range(stems))          # mixed activation pairs inside one launch
    eng = _engine(F=F, T=T, stem_modes=modes, variant=srt.VARIANT_VST, max_tiles=ntiles, impl=srt.IMPL_MFMA)
    for s in range(stems):
        eng.set_coeff(s, coeffs(s))
    x = _mag_input(oracle, ntiles, T, F, seed=500 + T + F)
    xd = torch.from_numpy(x).cuda()
    masks = eng.forward(xd).cpu().numpy()
    assert np.isfinite(masks).all()
    worst = (0.0, 0.0)
    for s in check_stems:
        for t in sorted({0, ntiles // 2, ntiles - 1}):
            worst = max(worst, _check_taps(eng, oracle, coeffs(s), x[t], modes[s], s, t, masks, "T=%d F=%d" % (T, F)))
    ks = _layer_kernels(eng, xd)
    for lvl in (1, 2, 3, 4, 5):
        name = "up%d" % lvl
        assert ks[name].startswith("srt_dec_wino") == _wino_expected(T, F, lvl), (name, ks[name], T >> (7 - lvl), F >> (7 - lvl))
    assert not ks["up6"].startswith("srt_dec_wino")
    for lvl in range(1, 7):                                  # the encoder's Winograd form (down3..down6 where the geometry fits); the bn+act copy of its first input comes from the direct layer in front
        name = "down%d" % lvl
        assert ks[name].startswith("srt_enc_wino32") == _enc_wino_expected(T, F, lvl), (name, ks[name])
    assert "actcopy" not in ks                                # the direct layer in front writes the bn+act copy itself (second output of srt_enc_mfma2)
    eng.close()
    print("wino odd geometry %dx%d x%d x%d: worst tap rel-rms %.3g, max-abs/peak %.3g; %s" % (
        T, F, ntiles, stems, worst[0], worst[1], {k: v for k, v in ks.items() if k.startswith("up")}))


@pytest.mark.parametrize("T,F,ntiles,stems", [
    (64, 512, 33, 4),       # up6 input 32 x 256: four full 64-pixel columns, 132 instances (odd tile count)
    (64, 576, 29, 4),       # 32 x 288: a 32-pixel last column (W % 64 != 0): the right image edge inside a DMA piece
    (128, 320, 43, 4),      # 64 x 160: three columns, the last one half empty; 32 chunks per column
])
def test_up6_streamed_form(oracle, coeffs, T, F, ntiles, stems):
    """Batches that fill the chip with 64-pixel column workgroups run up6 as srt_up6_stream_kernel (csrc/srt_nn.hip: LDS-DMA input, one
    workgroup streams down a column, ring of tap rows).  Every tensor of the first, an interior and the last tile of the first and the last
    stem against the oracle, at the tolerance of the other kernels; the engine must report the streamed kernel for up6."""
    import torch
    import spleeterrt_amd as srt
    modes = tuple((s + 1) % 2 for s in range(stems))
    eng = _engine(F=F, T=T, stem_modes=modes, variant=srt.VARIANT_VST, max_tiles=ntiles, impl=srt.IMPL_MFMA)
    for s in range(stems):
        eng.set_coeff(s, coeffs(s))
    x = _mag_input(oracle, ntiles, T, F, seed=900 + T + F)
    xd = torch.from_numpy(x).cuda()
    masks = eng.forward(xd).cpu().numpy()
    assert np.isfinite(masks).all()
    worst = (0.0, 0.0)
    for s in (0, stems - 1):
        for t in sorted({0, ntiles // 2, ntiles - 1}):
            worst = max(worst, _check_taps(eng, oracle, coeffs(s), x[t], modes[s], s, t, masks, "up6 stream T=%d F=%d" % (T, F)))
    ks = _layer_kernels(eng, xd)
    assert ks["up6"].startswith("srt_up6_stream_kernel"), ks["up6"]
    assert ks["up7"].startswith("srt_head_rows_kernel"), ks["up7"]      # the head with four output rows per thread (batches of >= 1024 of its workgroups)
    eng.close()
    print("up6 streamed %dx%d x%d x%d: worst tap rel-rms %.3g, max-abs/peak %.3g" % (T, F, ntiles, stems, worst[0], worst[1]))


@pytest.mark.parametrize("T,F,ntiles,stems,precision", [
    (64, 512, 96, 1, "f32"),        # one stem: half an M tile; 4 columns x 96 tiles x 2 runs of 4 intervals
    (64, 512, 96, 2, "f16"),        # two stems: one full M tile, fp16 storage (raw + act(BN(.)) as halves, epilogue constants from LDS)
    (128, 256, 192, 5, "f16"),      # BASELINE configs[4]'s five stems: four on the four-wave form, the fifth here; 2 columns x 192 tiles x 2 runs of 8 intervals
    (64, 512, 97, 5, "f32"),        # ... on fp32 tensors, odd tile count
])
def test_down1_streamed_two_wave_form(oracle, coeffs, T, F, ntiles, stems, precision):
    """Round 6: launches of one M tile (one or two stems, e.g. the fifth stem of five) run down1 as srt_down1_stream_kernel<.., NW = 2> (csrc/srt_nn2.hip: two waves per
    column workgroup, each column cut into two runs of intervals) instead of the tiled kernel.  Same MFMA chain per output: conv1 and the act1 tap must be BIT-IDENTICAL to the
    tiled form (SPLEETERRT_D1S2=0) on the first, an interior and the last tile - including the rows either side of the cut between the two runs - the engine must name the
    kernel, and the masks hold the mode's tolerance against the oracle."""
    import os
    import torch
    import spleeterrt_amd as srt
    modes = tuple((s + 1) % 2 for s in range(stems))
    kw = dict(precision=srt.PREC_F16) if precision == "f16" else dict(impl=srt.IMPL_MFMA)
    eng = _engine(F=F, T=T, stem_modes=modes, variant=srt.VARIANT_VST, max_tiles=ntiles, **kw)
    for s in range(stems):
        eng.set_coeff(s, coeffs(s))
    x = _mag_input(oracle, ntiles, T, F, seed=8800 + T + F + stems)
    xd = torch.from_numpy(x).cuda()
    last = stems - 1                                             # the stem(s) of the remainder group
    picks = [(last, t) for t in sorted({0, ntiles // 2, ntiles - 1})]
    old = os.environ.get("SPLEETERRT_D1S2")
    old_f16 = os.environ.get("SPLEETERRT_D1F16")
    os.environ["SPLEETERRT_D1F16"] = "0"                         # (fp16 mode: the fp32-MFMA streamed kernels this test is about, not srt_down1_f16_kernel)
    try:
        os.environ["SPLEETERRT_D1S2"] = "0"
        m0 = eng.forward(xd).cpu().numpy().copy()
        k0 = _layer_kernels(eng, xd)
        assert "srt_enc_mfma2<" in k0["down1"], k0["down1"]
        ref = {(n,) + st: eng.tensor(n, *st) for st in picks for n in ("conv1", "act1")}
        os.environ["SPLEETERRT_D1S2"] = "1"
        m1 = eng.forward(xd).cpu().numpy()
        k1 = _layer_kernels(eng, xd)
        assert "srt_down1_stream_kernel<0, %s, 2, " % ("true" if precision == "f16" else "false") in k1["down1"], k1["down1"]
        for key, want in ref.items():
            got = eng.tensor(key[0], key[1], key[2])
            assert np.array_equal(got, want), (key, int((got != want).sum()))
        assert np.array_equal(m0, m1)
    finally:
        if old is None:
            os.environ.pop("SPLEETERRT_D1S2", None)
        else:
            os.environ["SPLEETERRT_D1S2"] = old
        if old_f16 is None:
            os.environ.pop("SPLEETERRT_D1F16", None)
        else:
            os.environ["SPLEETERRT_D1F16"] = old_f16
    tol = 2e-2 if precision == "f16" else MASK_TOL_EXACT
    for s, t in picks:
        y = oracle.forward(coeffs(s), x[t], modes[s], oracle.VARIANT_VST)
        assert float(np.abs(m1[s, t] - y).max()) <= tol, (s, t)
    if precision == "f32":
        _check_taps(eng, oracle, coeffs(last), x[picks[1][1]], modes[last], last, picks[1][1], m1, "down1 two-wave")
    eng.close()


@pytest.mark.parametrize("T,F,ntiles,modes", [
    (64, 512, 96, (1, 0, 1, 0, 1)),        # BASELINE configs[4]'s five stems in ONE launch: three M tiles, the last half empty; 4 columns x 96 tiles, 8 intervals; both activations
    (128, 256, 192, (0, 1, 1)),            # three stems: two M tiles; 2 columns x 192 tiles, 16 intervals
    (64, 512, 97, (1,)),                   # one stem: half an M tile (4 stores per wave and interval), odd tile count
    (64, 1024, 48, (0, 1, 0, 1, 0, 1)),    # six stems: three full M tiles (24 stores per interval, the largest counted wait); 8 columns
])
def test_down1_fp16_mfma_form(oracle, coeffs, T, F, ntiles, modes):
    """Round 6: in the fp16 mode down1 runs on v_mfma_f32_32x32x16_f16 like every other conv layer (srt_down1_f16_kernel, csrc/srt_nn2.hip: k-group = the five taps of one
    (channel, ky) + three zero weights, every stem of the call in one launch, C8 outputs).  Not the fp32-MFMA chain: the magnitudes are rounded to halves first.  Checked:
    the engine names the kernel; conv1 of EVERY stem on the first, an interior and the last tile against the fp32 oracle convolution (<= 2 / 1024 of the tensor's peak:
    the input rounding + the output rounding) and against the fp32-MFMA streamed kernels (SPLEETERRT_D1F16=0) at the same bound; the act1 copy down2 reads against that
    form too; no value of the tiles is left unwritten (whole-tensor compare, NaN-filled first); masks within the mode's 2e-2 of the fp32 oracle."""
    import torch
    import spleeterrt_amd as srt
    stems = len(modes)
    eng = _engine(F=F, T=T, stem_modes=modes, variant=srt.VARIANT_VST, max_tiles=ntiles, precision=srt.PREC_F16)
    for s in range(stems):
        eng.set_coeff(s, coeffs(s))
    cf = coeffs
    x = _mag_input(oracle, ntiles, T, F, seed=9100 + T + F + stems)
    xd = torch.from_numpy(x).cuda()
    picks = sorted({0, ntiles // 2, ntiles - 1})
    with _env(SPLEETERRT_D1F16=0):
        m0 = eng.forward(xd).cpu().numpy().copy()
        k0 = _layer_kernels(eng, xd)
        assert "srt_down1_stream_kernel<0, true" in k0["down1"], k0["down1"]
        ref = {(n, s, t): eng.tensor(n, s, t) for s in range(stems) for t in picks for n in ("conv1", "act1")}
    with _env(SPLEETERRT_D1F16=1):
        m1 = eng.forward(xd).cpu().numpy()
        k1 = _layer_kernels(eng, xd)
        assert k1["down1"].startswith("srt_down1_f16_kernel<%d>" % ((stems + 1) // 2)), k1["down1"]
        assert k1["down2"].startswith("srt_enc_c8<"), k1["down2"]
        got = {k: eng.tensor(*k) for k in ref}
    lo = oracle.layout()
    worst = [0.0, 0.0, 0.0]
    for (n, s, t), g in got.items():
        want = ref[(n, s, t)]
        assert g.shape == want.shape and np.isfinite(g).all(), (n, s, t)
        e = float(np.abs(g - want).max() / np.abs(want).max())
        worst[0 if n == "conv1" else 1] = max(worst[0 if n == "conv1" else 1], e)
        assert e <= 2.0 / 1024, (n, s, t, e)
        if n == "conv1":
            c = cf(s)
            w = c[lo.down[0].w:lo.down[0].w + 25 * 2 * 16]
            o = oracle.conv5x5_s2(x[t], w, 16) + c[lo.down[0].b:lo.down[0].b + 16][:, None, None]
            e2 = float(np.abs(g - o).max() / np.abs(o).max())
            worst[2] = max(worst[2], e2)
            assert e2 <= 2.0 / 1024, (s, t, e2)
    for s in range(stems):
        for t in picks:
            y = oracle.forward(cf(s), x[t], modes[s], oracle.VARIANT_VST)
            assert float(np.abs(m1[s, t] - y).max()) <= 2e-2, (s, t)
            assert float(np.abs(m0[s, t] - y).max()) <= 2e-2, (s, t)
    print("down1 on the fp16 MFMA %dx%d x%d x%d stems: conv1 / act1 vs the fp32-MFMA form %.3g / %.3g of the peak, conv1 vs the oracle %.3g" % (T, F, ntiles, stems, worst[0], worst[1], worst[2]))
    eng.close()


@pytest.mark.parametrize("T,F,ntiles,stems,precision,variant", [
    (64, 512, 33, 4, "f32", "vst"),       # four full 64-pixel columns per instance, 132 instances; exact sigmoid
    (64, 576, 29, 4, "f32", "lut"),       # W = 288: a 32-pixel last column (the right image edge inside the halo the head needs); table sigmoid
    (128, 256, 65, 4, "f16", "vst"),      # fp16 activation storage (8-pixel DMA pieces; F % 256 == 0, so every column is full), two columns per instance, 34 intervals
    (256, 1024, 16, 4, "f16", "lut"),     # the bench tile geometry (8 columns x 131 intervals) on fp16 storage
    (256, 1024, 16, 5, "f32", "vst"),     # ... and on fp32 tensors, five stems (one workgroup per CU: 88.8 KB of LDS)
])
def test_up6_and_head_in_one_pass(oracle, coeffs, T, F, ntiles, stems, precision, variant):
    """srt_up6_head_kernel (csrc/srt_nn.hip) - the streamed up6 keeps its output rows in an LDS ring and emits the two mask planes itself, so the
    1-channel plane is neither written nor re-read.  Same MFMA chains, gather order and head FMA chain as srt_up6_stream_kernel + srt_head_rows_kernel:
    the masks of the whole batch must be BIT-IDENTICAL to the two-kernel form (SPLEETERRT_FUSE_HEAD=0), the engine must name the fused kernel for up6 and
    launch no up7, the up6 tap (re-materialised on demand by srtCopyTensor) must equal the two-kernel form's plane bit for bit, and the masks hold the
    mode's tolerance against the fp32 oracle on the first, an interior and the last tile of the first and the last stem."""
    import os
    import torch
    import spleeterrt_amd as srt
    modes = tuple((s + 1) % 2 for s in range(stems))
    var = srt.VARIANT_VST if variant == "vst" else srt.VARIANT_EXE
    ovar = oracle.VARIANT_VST if variant == "vst" else oracle.VARIANT_EXE
    kw = dict(precision=srt.PREC_F16) if precision == "f16" else dict(impl=srt.IMPL_MFMA)
    eng = _engine(F=F, T=T, stem_modes=modes, variant=var, max_tiles=ntiles, **kw)
    for s in range(stems):
        eng.set_coeff(s, coeffs(s))
    x = _mag_input(oracle, ntiles, T, F, seed=7700 + T + F)
    xd = torch.from_numpy(x).cuda()
    old = os.environ.get("SPLEETERRT_FUSE_HEAD")
    old_c8 = os.environ.get("SPLEETERRT_C8")
    try:
        os.environ["SPLEETERRT_C8"] = "0"                         # the one-pass form reads planar tensors only (fp16 storage: the planar kernels of srt_nn3.hip in front of it)
        os.environ["SPLEETERRT_FUSE_HEAD"] = "0"
        two = eng.forward(xd).cpu().numpy().copy()
        k2 = _layer_kernels(eng, xd)
        assert k2["up6"].startswith("srt_up6_stream_kernel<") and k2["up7"].startswith("srt_head_rows_kernel<"), (k2["up6"], k2.get("up7"))
        picks = [(s, t) for s in (0, stems - 1) for t in sorted({0, ntiles // 2, ntiles - 1})]
        planes = {st: eng.tensor("up6", *st) for st in picks}
        os.environ["SPLEETERRT_FUSE_HEAD"] = "1"
        eng.forward(xd.clone())                                  # (another batch in between would be better still: the plane buffer keeps the two-kernel run's values)
        one = eng.forward(xd).cpu().numpy()
        k1 = _layer_kernels(eng, xd)
        assert k1["up6"].startswith("srt_up6_head_kernel<64, 2, %s" % ("true" if precision == "f16" else "false")), k1["up6"]
        assert "up7" not in k1, k1
        assert np.array_equal(one, two), "masks differ: %d values, worst %g" % (int((one != two).sum()), float(np.abs(one - two).max()))
        for st in picks:
            assert np.array_equal(eng.tensor("up6", *st), planes[st]), st
        for _ in range(2):                                       # run-to-run stability of the one-pass form
            assert np.array_equal(eng.forward(xd).cpu().numpy(), one)
    finally:
        for k, v in (("SPLEETERRT_FUSE_HEAD", old), ("SPLEETERRT_C8", old_c8)):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    tol = 2e-2 if precision == "f16" else (MASK_TOL_EXACT if variant == "vst" else MASK_TOL_LUT)
    worst = 0.0
    for s, t in picks:
        ref = oracle.forward(coeffs(s), x[t], modes[s], ovar)
        worst = max(worst, float(np.abs(one[s, t] - ref).max()))
        assert worst <= tol, (s, t, worst)
    eng.close()
    print("up6 + head in one pass %dx%d x%d x%d %s %s: masks bit-identical to the two-kernel form, worst mask error vs the oracle %.3g" % (T, F, ntiles, stems, precision, variant, worst))


def test_up6_streamed_form_fp16_storage(oracle, coeffs):
    """The streamed up6 on fp16 activation tensors (srt_up6_stream_kernel<.., H16>, round 4: 8-pixel DMA pieces, the tiled fp16 kernel's two
    v_mfma_f32_32x32x16_f16 per 32 pixels): named by the engine at 33 tiles x 4 stems of 64 x 512, its output plane bit-identical to the tiled kernel's
    on the same tile evaluated alone, and the masks of the batch inside the fp16 tolerance of BASELINE configs[4] against the fp32 oracle."""
    import torch
    import spleeterrt_amd as srt
    T, F, stems, ntiles = 64, 512, 4, 33
    modes = (1, 0, 1, 0)
    eng = _engine(F=F, T=T, stem_modes=modes, variant=srt.VARIANT_VST, max_tiles=ntiles, precision=srt.PREC_F16)
    for s in range(stems):
        eng.set_coeff(s, coeffs(s))
    x = _mag_input(oracle, ntiles, T, F, seed=1616)
    xd = torch.from_numpy(x).cuda()
    masks = eng.forward(xd).cpu().numpy()
    ks = _layer_kernels(eng, xd)
    assert ks["up6"].startswith("srt_up6_stream_kernel<64, 2, 0, true"), ks["up6"]
    planes = {(s, t): eng.tensor("up6", s, t) for s in (0, 3) for t in (0, 17, ntiles - 1)}
    for (s, t), g in planes.items():
        ref = oracle.forward(coeffs(s), x[t], modes[s], oracle.VARIANT_VST)
        assert np.abs(masks[s, t] - ref).max() <= 2e-2, (s, t, float(np.abs(masks[s, t] - ref).max()))
    for t in (0, ntiles - 1):
        one = torch.from_numpy(x[t:t + 1]).cuda()
        eng.forward(one)
        k1 = _layer_kernels(eng, one)
        assert k1["up6"].startswith("srt_up6_kernel<"), k1["up6"]
        for s in (0, 3):
            assert np.array_equal(eng.tensor("up6", s, 0), planes[(s, t)]), (s, t)
    eng.close()


@pytest.mark.parametrize("T,F,ntiles,modes", [(64, 512, 9, (1, 0)), (128, 1024, 5, (1, 0, 1, 1)), (64, 256, 17, (0,)), (192, 768, 6, (1, 1, 0)), (256, 1024, 5, (0, 1, 1, 1))])   # the last: the bench tile geometry at an odd tile count (up2 = two instances of 8 x 32 per unit, the last group half empty)
def test_fp16_c8_layers(oracle, coeffs, T, F, ntiles, modes):
    """fp16 storage, launches above 16 instances (round 6, csrc/srt_nn5.hip): the tensors between down2 and up5 are channel-interleaved by eight and down3..down6 /
    up1..up5 run on the DMA-fed kernels.  Every tensor of sampled instances (first, middle, last tile: the last one sits in a partly empty instance group of the deep
    layers) against (a) the SAME tile evaluated alone, which takes the planar kernels of srt_nn3.hip (same products, same order: fp16 rounding noise at most), and
    (b) the fp32 CPU oracle at the fp16 tolerance class of BASELINE configs[4] (masks <= 2e-2).  Geometries: F = 256 (one-pixel-high deep layers, everything a
    partial tile), T = 192 / F = 768 (tile counts that are not powers of two), the bench's 1024 bins."""
    import torch
    import spleeterrt_amd as srt
    S = len(modes)
    assert S * ntiles > 16
    eng = _engine(F=F, T=T, stem_modes=modes, variant=srt.VARIANT_VST, max_tiles=ntiles, precision=srt.PREC_F16)
    for s in range(S):
        eng.set_coeff(s, coeffs(s))
    x = _mag_input(oracle, ntiles, T, F, seed=606)
    xd = torch.from_numpy(x).cuda()
    masks = eng.forward(xd).cpu().numpy()
    assert np.isfinite(masks).all()
    names = ["conv%d" % i for i in range(1, 7)] + ["act%d" % i for i in range(1, 6)] + ["up%d" % i for i in range(1, 7)]
    sample = [(s, t) for s in sorted({0, S - 1}) for t in sorted({0, ntiles // 2, ntiles - 1})]
    big = {(s, t): {n: eng.tensor(n, s, t) for n in names} for (s, t) in sample}
    ks = _layer_kernels(eng, xd)
    for n in ("down3", "down4", "down5", "down6"):
        assert ks[n].startswith("srt_enc_c8<"), (n, ks[n])
    for n in ("up1", "up2", "up3", "up4", "up5"):
        assert ks[n].startswith("srt_dec_c8<"), (n, ks[n])
    worst = 0.0
    for (s, t) in sample:
        ref = oracle.forward(coeffs(s), x[t], modes[s], oracle.VARIANT_VST)
        d = float(np.abs(masks[s, t] - ref).max())
        assert d <= 2e-2, (s, t, d)
    for t in sorted({t for (_, t) in sample}):
        one = torch.from_numpy(x[t:t + 1]).cuda()
        m1 = eng.forward(one).cpu().numpy()
        k1 = _layer_kernels(eng, one)
        assert k1["down3"].startswith("srt_enc_f16<") and k1["up2"].startswith("srt_dec_f16<"), k1
        for s in sorted({s for (s, _) in sample}):
            for n in names:
                a, b = big[(s, t)][n], eng.tensor(n, s, 0)
                err = float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
                worst = max(worst, err)
                assert err <= 2e-3, "%s stem %d tile %d: C8 batch vs planar single tile, max abs / peak %g at %r" % (
                    n, s, t, err, np.unravel_index(int(np.abs(a - b).argmax()), a.shape))
            assert float(np.abs(masks[s, t] - m1[s, 0]).max()) <= 2e-3
    print("fp16 C8 %dx%d x%d: worst tap difference to the planar path %.3g of the tensor peak" % (T, F, ntiles, worst))
    eng.close()


def test_forward_lut_variant(oracle, coeffs):
    import torch
    import spleeterrt_amd as srt
    T, F = 64, 512
    eng = _engine(F=F, T=T, stem_modes=(1,), variant=srt.VARIANT_EXE, max_tiles=2)
    eng.set_coeff(0, coeffs(2))
    x = _mag_input(oracle, 2, T, F, seed=99)
    masks = eng.forward(torch.from_numpy(x).cuda()).cpu().numpy()
    for t in range(2):
        y = oracle.forward(coeffs(2), x[t], 1, oracle.VARIANT_EXE)
        d = np.abs(masks[0, t] - y)
        assert d.max() <= MASK_TOL_LUT
        assert np.mean(d > MASK_TOL_EXACT) < 1e-3        # only points straddling the LUT's +-7 clip may exceed the tight bound
    eng.close()


def test_forward_lut_variant_large_batch(oracle, coeffs):
    """the LUT-sigmoid flavour (Executable) of the head's large-batch kernel (srt_head_rows_kernel<true, 4>): 140 instances of 64 x 512"""
    import torch
    import spleeterrt_amd as srt
    T, F, ntiles = 64, 512, 35
    eng = _engine(F=F, T=T, stem_modes=(1, 0, 1, 0), variant=srt.VARIANT_EXE, max_tiles=ntiles)
    for s in range(4):
        eng.set_coeff(s, coeffs(s))
    x = _mag_input(oracle, ntiles, T, F, seed=77)
    xd = torch.from_numpy(x).cuda()
    masks = eng.forward(xd).cpu().numpy()
    for s, t in ((0, 0), (1, 17), (3, 34)):
        y = oracle.forward(coeffs(s), x[t], (1, 0, 1, 0)[s], oracle.VARIANT_EXE)
        d = np.abs(masks[s, t] - y)
        assert d.max() <= MASK_TOL_LUT, (s, t, d.max())
        assert np.mean(d > MASK_TOL_EXACT) < 1e-3
    ks = _layer_kernels(eng, xd)
    assert ks["up7"].startswith("srt_head_rows_kernel<true"), ks["up7"]
    eng.close()


def test_fp16_container(oracle):
    import torch
    import spleeterrt_amd as srt
    T, F = 64, 512
    h = oracle.synth_coeff_fp16(3)
    # sprinkle half-denormals: the reference flushes them to zero (main.c:431)
    h = h.copy(); h[1000:1010] = np.arange(1, 11, dtype=np.uint16); h[2000] = 0x8001
    c = oracle.fp16_expand(h)
    e1 = _engine(F=F, T=T, stem_modes=(1,), variant=srt.VARIANT_VST, max_tiles=1)
    e2 = _engine(F=F, T=T, stem_modes=(1,), variant=srt.VARIANT_VST, max_tiles=1)
    e1.set_coeff_fp16(0, h)
    e2.set_coeff(0, c)
    x = torch.from_numpy(_mag_input(oracle, 1, T, F, seed=5)).cuda()
    assert torch.equal(e1.forward(x), e2.forward(x))      # identical weights -> bit-identical masks
    e1.close(); e2.close()


def half_patterns_expected():
    """f32Decompress (Executable/main.c:423-434) worked out by hand for every one of the 65 536 half patterns, without the oracle's C:
    exponent 0 -> signed zero (denormals flushed); exponent 1..30 -> the IEEE value (numpy's own half -> float conversion); exponent 31 is NOT
    special-cased by the reference: the bits are re-biased like a normal number, giving +-65536 * (1 + m/1024) instead of Inf / NaN."""
    h = np.arange(65536, dtype=np.uint32)
    sign, ex, man = h >> 15, (h >> 10) & 31, h & 1023
    with np.errstate(all="ignore"):
        val = h.astype(np.uint16).view(np.float16).astype(np.float32)
    val = np.where(ex == 0, np.where(sign == 1, np.float32(-0.0), np.float32(0.0)), val)
    top = (65536.0 * (1.0 + man / 1024.0)).astype(np.float32)
    val = np.where(ex == 31, np.where(sign == 1, -top, top), val).astype(np.float32)
    return val


def test_fp16_expand_all_65536_patterns(oracle):
    """VERDICT r4 missing #3: the GPU expand kernel, the oracle's C restatement and the hand-derived table agree BIT FOR BIT on every half
    pattern (Inf/NaN patterns included - the reference does not special-case them).  main.c:423-434 itself cannot be built here (model.c absent)."""
    import spleeterrt_amd as srt
    want = half_patterns_expected()
    allh = np.arange(65536, dtype=np.uint16)
    assert np.array_equal(oracle.fp16_expand(allh).view(np.uint32), want.view(np.uint32))
    n = 9822725
    rng = np.random.RandomState(7)
    h = np.concatenate([allh, rng.permutation(allh), rng.randint(0, 65536, n - 2 * 65536).astype(np.uint16)])       # every pattern at least twice, at aligned and odd positions
    eng = _engine(F=512, T=64, stem_modes=(1,), variant=srt.VARIANT_VST, max_tiles=1)
    eng.set_coeff_fp16(0, h)
    got = eng.get_coeff(0)
    eng.close()
    assert np.array_equal(got.view(np.uint32), want[h].view(np.uint32))
    assert np.array_equal(got.view(np.uint32), oracle.fp16_expand(h).view(np.uint32))


def test_stft_matches_oracle(oracle):
    import torch
    n = 4096 * 6 + 8192 + 1500                               # ragged tail: last frame is zero padded
    L, R = oracle.synth_audio(n, 777, True)
    eng = _engine(F=1024, T=64, stem_modes=(1,), max_tiles=2)
    spec, mag = eng.stft(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda())
    spec = spec.cpu().numpy(); mag = mag.cpu().numpy()
    re, im = oracle.stft(L, R)
    rows = re.shape[1]
    peak = np.abs(re).max()
    assert spec.shape[1] == rows
    assert np.abs(spec[:, :, :2049, 0] - re[:, :, :2049]).max() <= 2e-6 * peak
    assert np.abs(spec[:, :, :2049, 1] - im[:, :, :2049]).max() <= 2e-6 * peak
    assert np.all(spec[:, :, 2049:, :] == 0)
    frames = eng.L.srtStftFrames(n)
    assert np.all(spec[:, frames:, :, :] == 0)              # rows the reference leaves calloc'ed
    for t in range(mag.shape[0]):
        ref = oracle.magnitude_tile(re, im, t * 64, 64, 1024)
        assert np.abs(mag[t] - ref).max() <= 2e-6 * np.abs(ref).max()
    eng.close()


def test_istft_roundtrip_and_oracle(oracle):
    import torch
    n = 4096 * 4 + 8192
    L, R = oracle.synth_audio(n, 31337, True)
    eng = _engine(F=512, T=64, stem_modes=(1, 1), oob_weights=(1.0, 0.25), max_tiles=1)
    Ld, Rd = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    spec, _ = eng.stft(Ld, Rd, want_mag=False)
    out = eng.istft(spec, None).cpu().numpy()               # all-ones mask
    re, im = oracle.stft(L, R)
    ref0 = oracle.istft(re, im)
    peak = np.abs(ref0).max()
    assert out.shape == (2, 2, ref0.shape[1])
    assert np.abs(out[0] - ref0).max() <= 2e-6 * peak       # stem 0: oob weight 1 -> plain istft
    # interior of the round trip reproduces the input (OLA gain 1): stftFix.c windows, SURVEY §4
    x = np.stack([L, R])
    assert np.abs(out[0][:, 4096:n - 4096] - x[:, 4096:n - 4096]).max() <= 1e-5
    # stem 1: bins >= F scaled by 0.25 (mask is all-ones below F)
    re2, im2 = re.copy(), im.copy()
    re2[:, :, 512:2049] *= 0.25; im2[:, :, 512:2049] *= 0.25
    ref1 = oracle.istft(re2, im2)
    assert np.abs(out[1] - ref1).max() <= 2e-6 * peak
    eng.close()


def test_istft_three_per_cu_kernel_against_the_table_kernel(oracle):
    """srt_istft_ola3_kernel (F <= 1024, three workgroups per CU) rebuilds the synthesis window per frame from two registers (1/3 - cos/3 by angle
    addition) where srt_istft_ola_kernel (F > 1024) reads the postWin table that reproduces InitSTFT's rounding (stftFix.c:329-341).  The two windows differ by a few
    1e-8 absolute (large only relative to the ~1e-7 taps at the window's edges).  Bounded here on identical input: the same spectrum, unit masks, out-of-band weight 1
    (so the band limit F drops out) through an F = 1024 engine (three-per-CU kernel) and an F = 1088 engine (table kernel): the outputs agree to 5e-7 of the peak
    (measured 3.1e-7: a few ulps), four times inside the tolerance either holds against the oracle."""
    import torch
    n = 4096 * 6 + 8192
    L, R = oracle.synth_audio(n, 4242, True)
    outs = []
    for F in (1024, 1088):
        eng = _engine(F=F, T=64, stem_modes=(1,), oob_weights=(1.0,), max_tiles=1)
        Ld, Rd = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
        spec, _ = eng.stft(Ld, Rd, want_mag=False)
        outs.append(eng.istft(spec, None).cpu().numpy())
        eng.close()
    re, im = oracle.stft(L, R)
    ref = oracle.istft(re, im)
    peak = float(np.abs(ref).max())
    d = float(np.abs(outs[0] - outs[1]).max())
    print("istft: three-per-CU kernel vs table kernel max abs difference %.3g (peak %.3g); vs oracle %.3g / %.3g" % (
        d, peak, float(np.abs(outs[0][0] - ref).max()), float(np.abs(outs[1][0] - ref).max())))
    assert d <= 5e-7 * peak
    assert np.abs(outs[0][0] - ref).max() <= 2e-6 * peak and np.abs(outs[1][0] - ref).max() <= 2e-6 * peak


def test_separate_end_to_end(oracle, coeffs):
    """PCM -> stems against the oracle's stft -> processMT -> istft (main.c:776-785), 2 stems, ragged tail tile."""
    import torch
    import spleeterrt_amd as srt
    T, F = 64, 512
    n = 4096 * 24 + 8192                                     # 104 rows -> 1 full tile + 40-row tail
    L, R = oracle.synth_audio(n, 777, True)
    modes = (1, 0)
    eng = _engine(F=F, T=T, stem_modes=modes, variant=srt.VARIANT_VST, max_tiles=2)
    for s in range(2):
        eng.set_coeff(s, coeffs(s))
    out = eng.separate(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()).cpu().numpy()
    re, im = oracle.stft(L, R)
    for s in range(2):
        r, i = re.copy(), im.copy()
        oracle.process_spectrogram(coeffs(s), r, i, F, T, modes[s], oracle.VARIANT_VST, 0.1)
        ref = oracle.istft(r, i)
        peak = np.abs(ref).max()
        assert _rel_rms(out[s], ref) <= 1e-4, "stem %d rel rms %g" % (s, _rel_rms(out[s], ref))
        assert np.abs(out[s] - ref).max() <= 1e-4 * peak
    eng.close()


def test_chunked_stream_equals_single_batch(oracle, coeffs):
    """Tile-range chunking (spleeterrt_amd.stream: what long streams and multi-GPU shards use) on the real engine:
    a 5-tile stream run as one batch == the same stream run as chunks of 2 tiles (+ halo) and stitched."""
    import torch
    import spleeterrt_amd as srt
    from spleeterrt_amd import stream
    T, F = 64, 512
    n = 4096 * 70 + 8192 + 700                                # 289 rows -> 5 tiles, ragged
    L, R = oracle.synth_audio(n, 99, True)
    Ld, Rd = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    big = _engine(F=F, T=T, stem_modes=(1, 0), variant=srt.VARIANT_VST, max_tiles=8)
    small = _engine(F=F, T=T, stem_modes=(1, 0), variant=srt.VARIANT_VST, max_tiles=2)
    for e in (big, small):
        for s in range(2):
            e.set_coeff(s, coeffs(s))
    ref = big.separate(Ld, Rd).cpu().numpy()
    for world in (1, 2):
        parts = []
        for rank in range(world):
            parts += stream.separate_stream(small, Ld, Rd, rank, world)
        got = stream.stitch(parts, n, 2)
        assert got.shape == ref.shape
        # OLA order at the seams + (small batches only) the split-K association of the conv sums
        assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max(), "world %d" % world
    big.close(); small.close()


@pytest.mark.parametrize("max_tiles", [1, 2, 8])
def test_host_stream_pipeline_equals_single_batch(oracle, coeffs, max_tiles):
    """srtSeparateHostStream (native chunking, copies overlapped with compute on three streams, overlaps carried on the
    device) == one resident batch, for chunk sizes that divide the stream unevenly, evenly, and not at all."""
    import torch
    import spleeterrt_amd as srt
    T, F = 64, 512
    n = 4096 * 70 + 8192 + 700                                # 289 rows -> 5 tiles, ragged
    L, R = oracle.synth_audio(n, 99, True)
    big = _engine(F=F, T=T, stem_modes=(1, 0), variant=srt.VARIANT_VST, max_tiles=8)
    eng = _engine(F=F, T=T, stem_modes=(1, 0), variant=srt.VARIANT_VST, max_tiles=max_tiles)
    for e in (big, eng):
        for s in range(2):
            e.set_coeff(s, coeffs(s))
    ref = big.separate(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()).cpu().numpy()
    got = eng.separate_host_stream(L, R)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()     # OLA order at chunk seams + split-K association at these tiny batches
    again = eng.separate_host_stream(L, R)                   # reusable, deterministic
    assert np.array_equal(got, again)
    big.close(); eng.close()


@pytest.mark.parametrize("T,F,stems", [(256, 1536, 1), (64, 576, 2), (128, 2048, 1), (64, 64, 1), (256, 1024, 5)])
def test_forward_other_geometries(oracle, coeffs, T, F, stems):
    """Geometries the reference is used with: the VST default (F=1536, T=256, PluginProcessor.cpp:124), the CLI's clamp
    range 512..2048 (main.c:741-748), a width whose deep levels are not multiples of 4 (falls back to the scalar-staging
    kernels), the smallest legal tile, and the 5-stem configuration (BASELINE configs[4], fp32 here)."""
    import torch
    import spleeterrt_amd as srt
    eng = _engine(F=F, T=T, stem_modes=(1,) * stems, variant=srt.VARIANT_VST, max_tiles=1)
    for s in range(stems):
        eng.set_coeff(s, coeffs(s))
    x = _mag_input(oracle, 1, T, F, seed=1000 + F)
    masks = eng.forward(torch.from_numpy(x).cuda()).cpu().numpy()
    check = range(stems) if T * F <= 64 * 1024 else (stems - 1,)          # the oracle needs seconds per large tile
    for s in check:
        y = oracle.forward(coeffs(s), x[0], 1, oracle.VARIANT_VST)
        d = np.abs(masks[s, 0] - y).max()
        assert d <= MASK_TOL_EXACT, "T=%d F=%d stem %d: max abs %g" % (T, F, s, d)
    eng.close()


@pytest.mark.parametrize("T,F,ntiles,modes", [
    (64, 128, 3, (1, 0, 1)),        # three tiles, mixed activation pairs in one launch
    (192, 320, 2, (0,)),            # T and F that are multiples of 64 but not powers of two
    (128, 192, 5, (1, 1)),          # odd tile count against the multi-instance tiles of down6 / up1 (NI = 4 / 2)
    (320, 64, 2, (0, 1)),           # tall, minimal width: every layer below down1 takes the scalar-staging fallback
    (64, 1984, 1, (1,)),            # widest non-power-of-two
])
def test_forward_geometry_sweep(oracle, coeffs, T, F, ntiles, modes):
    """Forward parity on shapes chosen to hit the tile-edge, instance-group and fallback paths of the kernels (partial
    spatial tiles, tiles-per-workgroup remainders, widths where W % 4 != 0 deep in the net)."""
    import torch
    import spleeterrt_amd as srt
    eng = _engine(F=F, T=T, stem_modes=modes, variant=srt.VARIANT_VST, max_tiles=ntiles)
    for s in range(len(modes)):
        eng.set_coeff(s, coeffs(s))
    x = _mag_input(oracle, ntiles, T, F, seed=77 + T + F)
    masks = eng.forward(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.isfinite(masks).all()
    for s, mode in enumerate(modes):
        for j in sorted({0, ntiles - 1}):
            y = oracle.forward(coeffs(s), x[j], mode, oracle.VARIANT_VST)
            d = np.abs(masks[s, j] - y).max()
            assert d <= MASK_TOL_EXACT, "T=%d F=%d stem %d tile %d: max abs %g" % (T, F, s, j, d)
    eng.close()


@pytest.mark.parametrize("n", [4096, 4097, 4096 + 1023, 8192 + 5])
def test_minimum_length_signals(oracle, coeffs, n):
    """Shortest inputs the reference accepts (one transform; below 4096 samples it underflows, stftFix.c:378):
    1 computed frame, the remaining rows stay zero, the single tile is almost entirely padding."""
    import torch
    import spleeterrt_amd as srt
    T, F = 64, 512
    L, R = oracle.synth_audio(n, 3, True)
    eng = _engine(F=F, T=T, stem_modes=(1,), variant=srt.VARIANT_VST, max_tiles=1)
    eng.set_coeff(0, coeffs(0))
    out = eng.separate(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()).cpu().numpy()
    re, im = oracle.stft(L, R)
    oracle.process_spectrogram(coeffs(0), re, im, F, T, 1, oracle.VARIANT_VST, 0.1)
    ref = oracle.istft(re, im)
    assert out.shape[2] == ref.shape[1]
    assert np.abs(out[0] - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-6)
    eng.close()


def test_rejects_bad_arguments():
    import torch
    import spleeterrt_amd as srt
    with pytest.raises(srt.EngineError):
        _engine(F=500, T=64, stem_modes=(1,))                 # F, T must be multiples of 64 (spleeter.c:113-119)
    eng = _engine(F=512, T=64, stem_modes=(1,), max_tiles=1)
    x = torch.zeros((1, 2, 64, 512), device="cuda")
    with pytest.raises(srt.EngineError):
        eng.forward(x)                                        # weights not set
    z = torch.zeros(1000, device="cuda")
    with pytest.raises(srt.EngineError):
        eng.separate(z, z)                                    # shorter than one frame
    big = torch.zeros(64 * 1024 * 3, device="cuda")
    with pytest.raises(srt.EngineError):
        eng.stft(big, big)                                    # more tiles than max_tiles
    eng.close()


@pytest.mark.parametrize("prec,mask_tol", [("f16", 2e-2), ("f16x2", MASK_TOL_EXACT)])
@pytest.mark.parametrize("T,F", [(64, 512), (128, 1024), (64, 576)])    # F = 576: not a multiple of 256 -> fp32 tensors, BN + activation applied by the consuming fp16 encoder
def test_fp16_mfma_variants(oracle, coeffs, prec, mask_tol, T, F):
    """BASELINE configs[4]: fp16 MFMA conv (fp32 accumulate, fp32 STFT/iSTFT) against the CPU fp32 oracle.
    f16   : activations rounded to fp16         -> mask max-abs <= 2e-2
    f16x2 : activations split hi+lo, weights are fp16-representable -> products exact, held to the fp32 tolerance."""
    import torch
    import spleeterrt_amd as srt
    modes = (1, 0)
    eng = _engine(F=F, T=T, stem_modes=modes, variant=srt.VARIANT_VST, max_tiles=2,
                  precision=srt.PREC_F16 if prec == "f16" else srt.PREC_F16X2)
    for s in range(2):
        eng.set_coeff(s, coeffs(s))
    x = _mag_input(oracle, 2, T, F, seed=77)
    masks = eng.forward(torch.from_numpy(x).cuda()).cpu().numpy()
    worst = 0.0
    for s in range(2):
        for t in range(2):
            y = oracle.forward(coeffs(s), x[t], modes[s], oracle.VARIANT_VST)
            worst = max(worst, float(np.abs(masks[s, t] - y).max()))
    print("fp16 variant %s %dx%d worst mask err %.3g" % (prec, T, F, worst))
    assert worst <= mask_tol
    eng.close()


def test_f16x2_refuses_weights_that_are_not_fp16_values(oracle, coeffs):
    """SRT_PREC_F16X2 promises the fp32 tolerance, which holds only while every conv weight IS an fp16 value (the Executable's container, main.c:423-443).  A raw
    fp32 blob with 24-bit mantissas (the VST's .dat files may hold those, PluginProcessor.cpp:47-61) is refused at srtSetCoeff* with a message instead of being rounded
    silently - through every entry form (host, device) - and the engine keeps refusing srtForward for that sub-network; the same blob is accepted in f32 and in f16 (whose
    tolerance class says so), and an fp16-representable blob is accepted in f16x2 and still meets the fp32 tolerance."""
    import torch
    import spleeterrt_amd as srt
    T, F = 64, 512
    good = coeffs(0)
    rng = np.random.default_rng(5)
    bad = (good * (1.0 + rng.uniform(-3e-4, 3e-4, good.shape))).astype(np.float32)          # 24-bit mantissas
    assert np.any(bad.astype(np.float16).astype(np.float32) != bad)
    eng = _engine(F=F, T=T, stem_modes=(1,), variant=srt.VARIANT_VST, max_tiles=1, precision=srt.PREC_F16X2)
    x = _mag_input(oracle, 1, T, F, seed=31)
    for setter in (lambda w: eng.set_coeff(0, w), lambda w: eng.set_coeff(0, torch.from_numpy(w).cuda())):
        with pytest.raises(srt.EngineError, match="fp16-representable"):
            setter(bad)
        with pytest.raises(srt.EngineError, match="weights not set"):
            eng.forward(torch.from_numpy(x).cuda())
    eng.set_coeff(0, good)                                    # a representable blob afterwards: accepted, fp32-level parity
    m = eng.forward(torch.from_numpy(x).cuda()).cpu().numpy()
    y = oracle.forward(good, x[0], 1, oracle.VARIANT_VST)
    assert float(np.abs(m[0, 0] - y).max()) <= MASK_TOL_EXACT
    eng.close()
    for prec, tol in ((srt.PREC_F32, MASK_TOL_EXACT), (srt.PREC_F16, 2e-2)):                # the other modes take the same blob
        e2 = _engine(F=F, T=T, stem_modes=(1,), variant=srt.VARIANT_VST, max_tiles=1, precision=prec)
        e2.set_coeff(0, bad)
        m = e2.forward(torch.from_numpy(x).cuda()).cpu().numpy()
        y = oracle.forward(bad, x[0], 1, oracle.VARIANT_VST)
        assert float(np.abs(m[0, 0] - y).max()) <= tol
        e2.close()


@pytest.mark.parametrize("prec,mask_tol,stem_tol", [("f16", 2e-2, 1e-2), ("f16x2", MASK_TOL_EXACT, 1e-4)])
def test_config4_five_stems_fp16_end_to_end(oracle, coeffs, prec, mask_tol, stem_tol):
    """BASELINE configs[4] as named: 5 stems (the fifth = one more spleeterCoeff blob), T=256, F=1024, fp16-MFMA conv
    with fp32 STFT / iSTFT, PCM -> stems, tolerance-checked against the CPU fp32 oracle: mask max-abs <= 2e-2 and stems rel-RMS <= 1e-2
    (BASELINE.md §4).  The split form (f16x2, exact products for fp16-representable weights) is held to the fp32 tolerances."""
    import torch
    import spleeterrt_amd as srt
    T, F, S = 256, 1024, 5
    n = T * 1024                                              # one tile: 256 rows, 253 transformed frames
    L, R = oracle.synth_audio(n, 2025, True)
    oob = (0.25, 0.0, 0.25, 0.25, 0.1)
    eng = _engine(F=F, T=T, stem_modes=(1,) * S, oob_weights=oob, variant=srt.VARIANT_VST, max_tiles=1,
                  precision=srt.PREC_F16 if prec == "f16" else srt.PREC_F16X2)
    for s in range(S):
        eng.set_coeff(s, coeffs(s))
    Ld, Rd = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    out = eng.separate(Ld, Rd).cpu().numpy()
    spec, mag = eng.stft(Ld, Rd)
    masks = eng.forward(mag).cpu().numpy()
    re, im = oracle.stft(L, R)
    assert re.shape[1] == T
    x = oracle.magnitude_tile(re, im, 0, T, F)
    worst_m = worst_s = 0.0
    for s in range(S):
        y = oracle.forward(coeffs(s), x, 1, oracle.VARIANT_VST)
        worst_m = max(worst_m, float(np.abs(masks[s, 0] - y).max()))
        r, i = re.copy(), im.copy()
        r[:, :, :F] *= y; i[:, :, :F] *= y                    # main.c:473-485 (one full tile: no tail rows)
        r[:, :, F:2049] *= np.float32(oob[s]); i[:, :, F:2049] *= np.float32(oob[s])    # main.c:486-493
        ref = oracle.istft(r, i)
        worst_s = max(worst_s, _rel_rms(out[s], ref))
    print("configs[4] %s: worst mask err %.3g, worst stem rel rms %.3g" % (prec, worst_m, worst_s))
    assert worst_m <= mask_tol and worst_s <= stem_tol
    eng.close()


@pytest.mark.parametrize("T,F,ntiles,stems,variant", [(256, 1024, 8, 2, "vst"), (64, 512, 33, 5, "lut"), (128, 1024, 11, 3, "vst")])
def test_fp16_mode_masks_as_halves_between_head_and_inverse(oracle, coeffs, T, F, ntiles, stems, variant):
    """Round 6: in the fp16 mode srtSeparate keeps ITS OWN mask buffer (head -> inverse transform, never handed to a caller) as halves where the head launch is
    large enough (srt_head_rows_kernel<.., 4, true> writes them, srt_istft_ola3_kernel<4, false, true> reads them): the same sigmoid values rounded once to 11 bits.
    Checked: the engine names both kernels; the stems equal the float-mask form (SPLEETERRT_M16=0) within 1e-3 of their peak and 5e-4 rel-RMS (a mask in [0, 1] rounded to
    a half is off by <= 2.5e-4 of itself); srtForward's masks - the caller's - stay floats and are untouched by the switch; a ratio-mask engine keeps float masks.
    (The float-mask chain itself is held against the CPU oracle by the end-to-end tests; the signal ends in a tail tile.)"""
    import torch
    import spleeterrt_amd as srt
    n = ntiles * T * 1024 - 3000                               # the last tile is a tail tile
    L, R = oracle.synth_audio(n, 77 + T + stems, True)
    modes = tuple((s + 1) % 2 for s in range(stems))
    var = srt.VARIANT_VST if variant == "vst" else srt.VARIANT_EXE
    eng = _engine(F=F, T=T, stem_modes=modes, variant=var, max_tiles=ntiles, precision=srt.PREC_F16)
    for s in range(stems):
        eng.set_coeff(s, coeffs(s))
    Ld, Rd = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()

    def kernels():
        eng.set_timing(True); eng.separate(Ld, Rd); ks = dict(eng.get_timing_kernels()); eng.set_timing(False)
        return ks
    with _env(SPLEETERRT_M16=0):
        ref = eng.separate(Ld, Rd).cpu().numpy().copy()
        k0 = kernels()
    assert k0["up7"].startswith("srt_head_rows_kernel<") and not k0["up7"].rstrip("> ").endswith("true"), k0["up7"]
    got = eng.separate(Ld, Rd).cpu().numpy()
    k1 = kernels()
    assert k1["up7"].startswith("srt_head_rows_kernel<") and k1["up7"].rstrip("> ").endswith(", 4, true"), k1["up7"]
    assert k1["istft"].startswith("srt_istft_ola3_kernel<4, false, true"), k1["istft"]
    assert np.isfinite(got).all()
    for s in range(stems):
        peak = float(np.abs(ref[s]).max())
        assert float(np.abs(got[s] - ref[s]).max()) <= 1e-3 * peak, (s, float(np.abs(got[s] - ref[s]).max()) / peak)
        assert _rel_rms(got[s], ref[s]) <= 5e-4, (s, _rel_rms(got[s], ref[s]))
    # the caller's masks are floats whatever the engine does inside srtSeparate
    spec, mag = eng.stft(Ld, Rd)
    m_a = eng.forward(mag).cpu().numpy()
    with _env(SPLEETERRT_M16=0):
        m_b = eng.forward(mag).cpu().numpy()
    assert m_a.dtype == np.float32 and np.array_equal(m_a, m_b)
    out2 = eng.istft(spec, torch.from_numpy(m_a).cuda()).cpu().numpy()          # the public inverse on float masks = the float-mask chain
    assert np.array_equal(out2, ref)
    eng.close()
    engr = _engine(F=F, T=T, stem_modes=modes, variant=var, max_tiles=ntiles, precision=srt.PREC_F16, ratio_mask=True)
    for s in range(stems):
        engr.set_coeff(s, coeffs(s))
    engr.set_timing(True); engr.separate(Ld, Rd); kr = dict(engr.get_timing_kernels()); engr.set_timing(False)
    assert not kr["up7"].rstrip("> ").endswith(", 4, true"), kr["up7"]
    engr.close()


@pytest.mark.parametrize("modes,ntiles,F", [((1, 0, 1, 1, 0), 2, 512), ((1, 0, 1, 1, 0, 1), 2, 512), ((1, 1, 0, 1, 1), 48, 1024)])
def test_more_than_four_stems_down1_groups(oracle, coeffs, modes, ntiles, F):
    """Five and six sub-networks in fp32 (round 5): down1 goes out as stacked groups of four stems + the remainder (4 + 1, 4 + 2), each group with its own
    slice of weights, bias, outputs and activation bits.  Mixed LeakyReLU / ELU stems so that a wrong shift of the activation mask shows; the 48-tile case
    puts the first group on the streamed down1 kernel (8 column strips x 48 tiles = 384 workgroups) and the remainder - one M tile - on its two-wave form
    (round 6: 8 x 48 x 2 runs = 768 workgroups of two waves).  Every tensor of the checked instances against the oracle (spleeter.c:182-300)."""
    import torch
    import spleeterrt_amd as srt
    T, S = 64, len(modes)
    eng = _engine(F=F, T=T, stem_modes=modes, variant=srt.VARIANT_VST, max_tiles=ntiles)
    cs = [coeffs(s) for s in range(S)]                           # seeds 2024 + stem: every sub-network has its own weights
    for s in range(S):
        eng.set_coeff(s, cs[s])
    x = _mag_input(oracle, ntiles, T, F, seed=909)
    masks = eng.forward(torch.from_numpy(x).cuda())
    kern = None
    if ntiles >= 48:
        eng.set_timing(True); eng.forward(torch.from_numpy(x).cuda(), masks); kern = [k for n, k in eng.get_timing_kernels() if n == "down1"]; eng.set_timing(False)
        assert len(kern) == 2 and kern[0].startswith("srt_down1_stream_kernel<0, false, 4,") and kern[1].startswith("srt_down1_stream_kernel<0, false, 2,"), kern
    masks = masks.cpu().numpy()
    for s, t in ((0, 0), (3, ntiles - 1), (4, 0), (S - 1, ntiles - 1), (1, ntiles // 2)):
        _check_taps(eng, oracle, cs[s], x[t], modes[s], s, t, masks=masks, tag="%d stems" % S)
    eng.close()


@pytest.mark.parametrize("stems", [2, 3])
def test_cli_flow_device_resident(oracle, coeffs, stems):
    """srtSeparateCli == the offline CLI's flow restated over the oracle (main.c:776-798 two outputs with the time-domain
    residual, :845-928 three outputs with the complex-domain residual chain), ragged tail tile included."""
    import torch
    import spleeterrt_amd as srt
    T, F = 64, 512
    n = 4096 * 24 + 8192
    L, R = oracle.synth_audio(n, 777, True)
    eng = _engine(F=F, T=T, stem_modes=(1, 0), variant=srt.VARIANT_VST, max_tiles=2)
    for s in range(2):
        eng.set_coeff(s, coeffs(s))
    out = eng.separate_cli(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda(), stems).cpu().numpy()
    ref = oracle.cli_separate(coeffs(0), coeffs(1), L, R, F, T, stems, oracle.VARIANT_VST, 0.1)
    assert out.shape == ref.shape
    peak = np.abs(ref).max()
    for k in range(stems):
        assert _rel_rms(out[k], ref[k]) <= 1e-4, "output %d rel rms %g" % (k, _rel_rms(out[k], ref[k]))
        assert np.abs(out[k] - ref[k]).max() <= 1e-4 * peak
    # the outputs of either flow add back up to the (reconstructed) input
    total = out.sum(axis=0)[:, 4096:n - 4096]
    assert np.abs(total[0] - L[4096:n - 4096]).max() <= 2e-5 and np.abs(total[1] - R[4096:n - 4096]).max() <= 2e-5
    eng.close()


@pytest.mark.parametrize("stems,minutes,max_tiles", [(3, 30.0, 16), (2, 2.0, 3), (3, 0.5, 1)])
def test_cli_flow_any_length_chunked_equals_resident(oracle, coeffs, stems, minutes, max_tiles):
    """VERDICT r2 missing #2 / next #7: the reference CLI walks its tiles one at a time over a host-resident spectrogram, so any file
    length works (main.c:455-495).  srtSeparateCliHost on an engine of max_tiles tiles walks a longer file chunk by chunk (uploads,
    residual chain and downloads overlapped; chunk seams added on the device BEFORE the time-domain subtraction) and must equal the
    one-resident-batch flow (itself oracle-tested in test_cli_flow_device_resident) to 2e-6 of the peak, for both output counts,
    chunk sizes that do and do not divide the file, down to one tile per chunk.  The first case is a 30-minute file at max_tiles = 16."""
    import spleeterrt_amd as srt
    T, F = 64, 512
    n = int(minutes * 60 * 44100) // 4096 * 4096 + 8192 + 333     # ragged tail tile
    rng = np.random.default_rng(1234)
    L = (rng.random(n, dtype=np.float32) - 0.5) * 0.2
    R = (rng.random(n, dtype=np.float32) - 0.5) * 0.2
    rows = (n + 1023) // 1024
    ntiles = (rows + T - 1) // T
    assert ntiles > max_tiles
    kw = dict(F=F, T=T, stem_modes=(1, 0), variant=srt.VARIANT_VST, batch_invariant=True)      # same kernels in both engines: what differs is the seams
    big = _engine(max_tiles=ntiles, **kw)
    for s in range(2):
        big.set_coeff(s, coeffs(s))
    ref = big.separate_cli_host(L, R, stems)
    big.close()
    eng = _engine(max_tiles=max_tiles, **kw)
    for s in range(2):
        eng.set_coeff(s, coeffs(s))
    got = eng.separate_cli_host(L, R, stems)
    assert got.shape == ref.shape and np.isfinite(got).all()
    peak = np.abs(ref).max()
    for k in range(stems):
        assert np.abs(got[k] - ref[k]).max() <= 2e-6 * peak, "output %d: %g of the peak" % (k, np.abs(got[k] - ref[k]).max() / peak)
    if minutes < 1:
        again = eng.separate_cli_host(L, R, stems)          # reusable, deterministic
        assert np.array_equal(got, again)
    eng.close()


def test_forward_stem_range_equals_full(oracle, coeffs):
    """srtForwardStems on [1,2) then [0,1) fills the same mask tensor as one srtForward over both sub-networks."""
    import torch
    import spleeterrt_amd as srt
    T, F = 64, 512
    x = torch.from_numpy(_mag_input(oracle, 2, T, F)).cuda()
    eng = _engine(F=F, T=T, stem_modes=(1, 0), variant=srt.VARIANT_VST, max_tiles=2)
    for s in range(2):
        eng.set_coeff(s, coeffs(s))
    full = eng.forward(x).clone()
    part = torch.zeros_like(full)
    eng.forward_stems(x, part, 1, 1)
    eng.forward_stems(x, part, 0, 1)
    # same arithmetic, but at these tiny batches the deep layers are split-K launches whose slice count follows the number of
    # instances in the launch: only the association of the K sums may differ
    assert float((full - part).abs().max()) <= 1e-4
    again = torch.zeros_like(full)
    eng.forward_stems(x, again, 1, 1)
    eng.forward_stems(x, again, 0, 1)
    assert torch.equal(part, again)                          # run to run: bit-stable (fixed reduction order)
    with pytest.raises(srt.EngineError):
        eng.forward_stems(x, part, 1, 2)
    eng.close()


def test_batch_invariant_switch(oracle, coeffs):
    """srt_config.batch_invariant: kernel choice by layer geometry only, no split-K - a tile's masks are bit-identical
    whatever the batch size, its slot, the stem range of the launch or the chunking, as on the reference's CPU path; and they still
    match the oracle.  The default mode (fastest kernel per launch size) is held to 1e-4 by the tests above."""
    import torch
    import spleeterrt_amd as srt
    from spleeterrt_amd import stream
    T, F, NT = 64, 512, 20
    x = _mag_input(oracle, NT, T, F, seed=2718)
    xd = torch.from_numpy(x).cuda()
    eng = _engine(F=F, T=T, stem_modes=(1, 0), variant=srt.VARIANT_VST, max_tiles=NT, batch_invariant=True)
    for s in range(2):
        eng.set_coeff(s, coeffs(s))
    full = eng.forward(xd).clone()                            # 40 instances
    for j in (0, 7, 19):
        alone = eng.forward(xd[j:j + 1].contiguous())         # 2 instances: the split-K regime of the default mode
        assert torch.equal(alone[:, 0], full[:, j]), "tile %d alone differs from the tile inside the batch" % j
    part = torch.zeros_like(full)
    eng.forward_stems(xd, part, 1, 1)
    eng.forward_stems(xd, part, 0, 1)
    assert torch.equal(part, full)
    three = eng.forward(xd[3:6].contiguous())                 # another batch size, other slots
    assert torch.equal(three, full[:, 3:6])
    eng.forward(xd)                                           # (srtCopyTensor reads the LAST forward's batch)
    _check_taps(eng, oracle, coeffs(1), x[7], 0, 1, 7, full.cpu().numpy(), "batch-invariant")
    ks = _layer_kernels(eng, xd[0:1].contiguous())
    assert all(ks[n].startswith("srt_dec_wino") for n in ("up3", "up4", "up5")), ks     # geometry decides, not the batch
    # audio: chunked == one batch up to the overlap-add association at the chunk seams only
    n = 4096 * 70 + 8192 + 700
    L, R = oracle.synth_audio(n, 99, True)
    Ld, Rd = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    small = _engine(F=F, T=T, stem_modes=(1, 0), variant=srt.VARIANT_VST, max_tiles=2, batch_invariant=True)
    for s in range(2):
        small.set_coeff(s, coeffs(s))
    ref = eng.separate(Ld, Rd).cpu().numpy()
    got = stream.stitch(stream.separate_stream(small, Ld, Rd, 0, 1), n, 2)
    assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max()
    got2 = small.separate_host_stream(L, R)
    assert np.abs(got2 - ref).max() <= 2e-6 * np.abs(ref).max()
    eng.close(); small.close()


def test_small_batch_split_k_and_graph_replay(oracle, coeffs):
    """The real-time regime (1 tile x 4 stems at the plugin's 256 x 1536, VST/Source/PluginProcessor.cpp:124): the deep layers run as
    split-K launches and the whole forward can be replayed as a hipGraph.  Masks vs the CPU oracle, bit-stable run to run,
    graph replay == eager."""
    import torch
    import spleeterrt_amd as srt
    T, F, S = 256, 1536, 4
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        eng = _engine(F=F, T=T, stem_modes=(1,) * S, variant=srt.VARIANT_VST, max_tiles=1)
        for s in range(S):
            eng.set_coeff(s, coeffs(s))
        x = _mag_input(oracle, 1, T, F, seed=31)
        xd = torch.from_numpy(x).cuda()
        m1 = eng.forward(xd).clone()
        m2 = eng.forward(xd).clone()
        assert torch.equal(m1, m2)
        eng.set_graph_mode(True)
        out = torch.empty_like(m1)
        eng.forward(xd, out)                                 # captures
        first = out.clone()
        out.zero_()
        eng.forward(xd, out)                                 # replays
        side.synchronize()
        assert torch.equal(first, m1) and torch.equal(out, m1)
        eng.set_graph_mode(False)
        eng.close()
    for s in (0, 3):
        y = oracle.forward(coeffs(s), x[0], 1, oracle.VARIANT_VST)
        assert np.abs(m1[s, 0].cpu().numpy() - y).max() <= MASK_TOL_EXACT


def test_ratio_mask(oracle, coeffs):
    """Optional cross-stem ratio mask: kernel vs the numpy restatement, and the end-to-end switch."""
    import torch
    import spleeterrt_amd as srt
    T, F = 64, 512
    S = 3
    m = (oracle.lcg(99, S * 2 * 2 * T * F, 1.0) + 0.5).reshape(S, 2, 2, T, F).astype(np.float32)
    m[0, 0, 0, 0, :8] = 0.0
    m[1, 0, 0, 0, :8] = 0.0
    m[2, 0, 0, 0, :4] = 0.0                                  # all-zero bins: eps keeps them finite (1/S each)
    eng = _engine(F=F, T=T, stem_modes=(1, 1, 1), variant=srt.VARIANT_VST, max_tiles=2)
    got = eng.ratio_mask(torch.from_numpy(m).cuda()).cpu().numpy()
    ref = oracle.ratio_mask(m)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 2e-7
    assert np.abs(got.sum(axis=0) - 1.0).max() <= 1e-6
    eng.close()

    n = 4096 * 12 + 8192
    L, R = oracle.synth_audio(n, 31, True)
    eng = _engine(F=F, T=T, stem_modes=(1, 0), variant=srt.VARIANT_VST, max_tiles=1, ratio_mask=True)
    for s in range(2):
        eng.set_coeff(s, coeffs(s))
    Ld, Rd = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    out = eng.separate(Ld, Rd).cpu().numpy()
    spec, mag = eng.stft(Ld, Rd)
    masks = eng.ratio_mask(eng.forward(mag))
    ref = eng.istft(spec, masks).cpu().numpy()
    # srtSeparate applies the ratio inside the inverse kernel's prologue (srt_ratio_of), this path through srt_ratio_mask_kernel: same operations, same bits
    bad = np.argwhere(out != ref)
    assert bad.size == 0, "%d samples differ, first at %r: %r vs %r (max abs %g)" % (len(bad), tuple(bad[0]), out[tuple(bad[0])], ref[tuple(bad[0])], np.abs(out - ref).max())
    eng.close()


def test_full_size_batch_properties(oracle, coeffs):
    """BASELINE configs[2] at full size (4 stems x 64 tiles of 256x1024), through size-independent properties:
    a T-periodic signal gives identical masks in every interior tile; tiles are independent - checked on a batch of 64 DISTINCT tiles
    (seeded noise per tile, so a cross-tile mix-up cannot hide behind identical inputs): permuting the tiles of the batch permutes the
    masks bit for bit, and a tile's masks do not depend on its batch mates; masks are finite and inside [0,1], unit masks give back the
    input (STFT -> iSTFT identity, stftFix.c round trip), and one (tile, stem) of the batch matches the CPU oracle.  (Per-tap parity at
    this launch size: test_shipped_launch_shapes_per_tap.)"""
    import torch
    import spleeterrt_amd as srt
    T, F, S, NT = 256, 1024, 4, 64
    eng = _engine(F=F, T=T, stem_modes=(1,) * S, oob_weights=(0.25, 0.0, 0.25, 0.25), variant=srt.VARIANT_VST, max_tiles=NT)
    for s in range(S):
        eng.set_coeff(s, coeffs(s))
    n = NT * T * 1024
    per = T * 1024
    L1, R1 = oracle.synth_audio(per, 777, True)
    L = torch.from_numpy(np.tile(L1, NT)).cuda()
    R = torch.from_numpy(np.tile(R1, NT)).cuda()
    spec, mag = eng.stft(L, R)
    assert tuple(mag.shape) == (NT, 2, T, F)
    masks = eng.forward(mag)
    assert torch.isfinite(masks).all() and float(masks.min()) >= 0.0 and float(masks.max()) <= 1.0
    # interior tiles of a T-periodic signal see identical magnitudes -> bit-identical masks, whatever their slot
    assert torch.equal(mag[5], mag[40])
    for j in (6, 31, 62):
        assert torch.equal(masks[:, 5], masks[:, j]), "tile %d differs from tile 5" % j
    # independence on DISTINCT tiles: a permutation of the batch permutes the masks, bit for bit (same kernels, same launch size; a tile that
    # read a neighbour's rows, channels or slot would change with its neighbours)
    xd = torch.from_numpy(_mag_input(oracle, NT, T, F, seed=9001)).cuda()
    assert not torch.equal(xd[5], xd[40])
    m1 = eng.forward(xd).clone()
    perm = torch.from_numpy(np.random.RandomState(7).permutation(NT)).cuda()
    m2 = eng.forward(xd[perm].contiguous())
    assert torch.equal(m2, m1[:, perm]), "masks of a permuted batch are not the permuted masks"
    xz = xd.clone(); xz[:31] = 0.0; xz[32:] = 0.0                        # tile 31 among silent batch mates
    assert torch.equal(eng.forward(xz)[:, 31], m1[:, 31])
    del m1, m2, xz
    # a tile evaluated alone (batch of 1, slot 0) == the same tile inside the 64-tile batch; alone, its deep layers run as
    # split-K launches (small-batch path), so only the association of the K sums differs
    alone = eng.forward(mag[17:18].contiguous())
    assert float((alone[:, 0] - masks[:, 17]).abs().max()) <= 1e-4
    assert torch.equal(alone, eng.forward(mag[17:18].contiguous()))      # bit-stable run to run
    # one (tile, stem) against the CPU oracle at the full tile size
    ref = oracle.forward(coeffs(2), mag[5].cpu().numpy(), 1, oracle.VARIANT_VST)
    assert np.abs(masks[2, 5].cpu().numpy() - ref).max() <= MASK_TOL_EXACT
    # unit masks: the inverse transform + overlap-add returns the input (away from the first / last three hops)
    rt = _engine(F=F, T=T, stem_modes=(1,), oob_weights=(1.0,), variant=srt.VARIANT_VST, max_tiles=NT)   # bins >= F untouched too
    back = rt.istft(spec, None)[0]
    rt.close()
    err = (back[:, 4096:n - 4096] - torch.stack([L, R])[:, 4096:n - 4096]).abs().max()
    assert float(err) <= 2e-6
    # end to end at full size: separate() == its three stages
    out = eng.separate(L, R)
    ref_out = eng.istft(spec, masks)
    assert torch.equal(out, ref_out)
    eng.close()


def test_config4_full_size_batch_properties_and_taps(oracle, coeffs):
    """BASELINE configs[4] exactly as `bench.py --stems 5 --precision f16` launches it - 64 tiles x 5 stems of 256 x 1024, fp16-MFMA conv with the mid-network tensors
    channel-interleaved by eight on the DMA-fed kernels of csrc/srt_nn5.hip (their grids, units per workgroup, the 4 + 1 down1 stem groups of this size).  Size-independent
    properties on 64 DISTINCT tiles: masks finite and inside [0, 1]; a permutation of the batch permutes the masks bit for bit and a tile among silent batch mates keeps its
    bits (tiles are independent: main.c:455-495); run to run bit-stable.  Then every tensor of three (stem, tile) instances - first / last stem, a tile in the middle and the
    last one - against the same tile evaluated alone on the planar kernels (<= 2e-3 of the tensor's peak) and their masks against the fp32 CPU oracle (<= 2e-2)."""
    import torch
    import spleeterrt_amd as srt
    T, F, S, NT = 256, 1024, 5, 64
    eng = _engine(F=F, T=T, stem_modes=(1,) * S, variant=srt.VARIANT_VST, max_tiles=NT, precision=srt.PREC_F16)
    for s in range(S):
        eng.set_coeff(s, coeffs(s))
    x = _mag_input(oracle, NT, T, F, seed=6006)
    xd = torch.from_numpy(x).cuda()
    m1 = eng.forward(xd).clone()
    assert torch.isfinite(m1).all() and float(m1.min()) >= 0.0 and float(m1.max()) <= 1.0
    ks = _layer_kernels(eng, xd)
    assert ks["down3"].startswith("srt_enc_c8<") and ks["down6"].startswith("srt_enc_c8<") and ks["up1"].startswith("srt_dec_c8<") and ks["up5"].startswith("srt_dec_c8<"), ks
    assert torch.equal(eng.forward(xd), m1)                                                      # bit-stable run to run
    names = ["conv%d" % i for i in range(1, 7)] + ["up%d" % i for i in range(1, 7)]
    sample = [(0, 33), (S - 1, 33), (S - 1, NT - 1)]                                             # (stem 4 = the down1 launch group of one)
    taps = {(s, t): {n: eng.tensor(n, s, t) for n in names} for (s, t) in sample}
    perm = torch.from_numpy(np.random.RandomState(11).permutation(NT)).cuda()
    m2 = eng.forward(xd[perm].contiguous())
    assert torch.equal(m2, m1[:, perm]), "masks of a permuted batch are not the permuted masks"
    xz = xd.clone(); xz[:33] = 0.0; xz[34:] = 0.0                                                # tile 33 among silent batch mates
    assert torch.equal(eng.forward(xz)[:, 33], m1[:, 33])
    del m2, xz
    for (s, t) in sample:
        ref = oracle.forward(coeffs(s), x[t], 1, oracle.VARIANT_VST)
        d = float(np.abs(m1[s, t].cpu().numpy() - ref).max())
        assert d <= 2e-2, (s, t, d)
    worst = 0.0
    for t in sorted({t for (_, t) in sample}):
        one = torch.from_numpy(x[t:t + 1]).cuda()
        eng.forward(one)                                                                         # 5 instances: the planar kernels
        for s in sorted({s for (s, tt) in sample if tt == t}):
            for n in names:
                a, b = taps[(s, t)][n], eng.tensor(n, s, 0)
                err = float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
                worst = max(worst, err)
                assert err <= 2e-3, "%s stem %d tile %d: C8 batch vs planar single tile, max abs / peak %g" % (n, s, t, err)
    print("configs[4] full size: worst tap difference between the C8 batch and the planar single tile %.3g of the tensor peak" % worst)
    eng.close()


# ------------------------------------------------------------------ the launch shapes the headline is timed on, per tap
SHIPPED_KERNELS = {      # bench.py's `layer_kernels` at 64 tiles x 4 stems of 256 x 1024 (profiles/r0x_bench_n1.json); template arguments may move with tuning, the families may not
    "down1": "srt_down1_stream_kernel<", "down2": "srt_enc_mfma2<", "down3": "srt_enc_wino32<", "down4": "srt_enc_wino32<", "down5": "srt_enc_wino32<", "down6": "srt_enc_wino32<",
    "up1": "srt_dec_wino32<", "up2": "srt_dec_wino32<", "up3": "srt_dec_wino32<", "up4": "srt_dec_wino32<", "up5": "srt_dec_wino", "up6": "srt_up6_stream_kernel<", "up7": "srt_head_rows_kernel<",
}


def test_shipped_launch_shapes_per_tap(oracle, coeffs):
    """BASELINE configs[2] exactly as bench.py launches it - 64 tiles x 4 stems of 256 x 1024, i.e. the grids, units per workgroup, column
    walks and kernel choices of the headline number - on 64 DISTINCT tiles (seeded noise per tile): every intermediate tensor of the first,
    an interior and the last tile of stems 0 and 3 against the CPU oracle (rel-RMS and max-abs / peak), the masks of those instances, the
    kernel families named by the engine, and the stems of two tiles end to end (PCM -> STFT -> masks -> iSTFT) against
    oracle.stft -> process_spectrogram -> istft."""
    import torch
    import spleeterrt_amd as srt
    T, F, S, NT = 256, 1024, 4, 64
    modes, oob = (1, 1, 1, 1), (0.25, 0.0, 0.25, 0.25)
    eng = _engine(F=F, T=T, stem_modes=modes, oob_weights=oob, variant=srt.VARIANT_VST, max_tiles=NT)
    for s in range(S):
        eng.set_coeff(s, coeffs(s))
    x = _mag_input(oracle, NT, T, F, seed=64256)
    xd = torch.from_numpy(x).cuda()
    masks = eng.forward(xd).cpu().numpy()
    worst = (0.0, 0.0)
    for s in (0, 3):
        for t in (0, 37, NT - 1):
            worst = max(worst, _check_taps(eng, oracle, coeffs(s), x[t], modes[s], s, t, masks, "64x4 256x1024"))
    ks = _layer_kernels(eng, xd)
    for name, fam in SHIPPED_KERNELS.items():
        assert ks[name].startswith(fam), (name, ks[name])
    assert "actcopy" not in ks
    # end to end on distinct audio: 64 tiles of seeded noise + tones, tiles 0 and 41 of every stem against the oracle chain of their own PCM span
    n = NT * T * 1024
    L, R = oracle.synth_audio(n, 20240, True)
    out = eng.separate(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()).cpu().numpy()
    for t in (0, 41):
        s0, s1 = t * T * 1024, (t + 1) * T * 1024 + 3072
        re, im = oracle.stft(L[s0:s1].copy(), R[s0:s1].copy())
        re, im = re[:, :T], im[:, :T]                                        # the tile's own rows (the halo frames belong to the next tile)
        lo, hi = s0 + 3072, s0 + T * 1024                                    # samples that only this tile's frames reach
        for s in range(S):
            r, i = re.copy(), im.copy()
            oracle.process_spectrogram(coeffs(s), r, i, F, T, modes[s], oracle.VARIANT_VST, oob[s])
            ref = oracle.istft(r, i)[:, 3072:T * 1024]
            got = out[s][:, lo:hi]
            assert _rel_rms(got, ref) <= 1e-4, (t, s, _rel_rms(got, ref))
            assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max(), (t, s)
    eng.close()
    print("shipped shapes 64x4 256x1024: worst tap rel-rms %.3g, max-abs/peak %.3g; %s" % (worst[0], worst[1], ks))


@pytest.mark.parametrize("T,F,ntiles,stems,taps", [
    (512, 1024, 1, 4, ((0, 0), (3, 0))),               # the reference CLI's DEFAULT geometry (main.c:701-702, README "official setting"), one tile: direct kernels + split-K
    (512, 1024, 5, 4, ((0, 0), (1, 2), (3, 4))),       # ... and as a 20-instance batch: the Winograd kernels at H up to 256
    (256, 2048, 5, 4, ((0, 0), (2, 2), (3, 4))),       # the CLI's upper clamp F = 2048 (main.c:745-748) above 16 instances: the Winograd kernels at W up to 1024
])
def test_reference_default_and_widest_geometries(oracle, coeffs, T, F, ntiles, stems, taps):
    """The geometries the reference ships with that no other test reaches: timeStep 512 x analyseBinLimit 1024 (its default) and the
    widest tile it accepts, 2048 bins, at batch sizes that take the Winograd kernels.  Every tensor of the listed (stem, tile) instances
    against the oracle, and the kernel families where the geometry fits."""
    import torch
    import spleeterrt_amd as srt
    modes = tuple((s + 1) % 2 for s in range(stems))
    eng = _engine(F=F, T=T, stem_modes=modes, variant=srt.VARIANT_VST, max_tiles=ntiles)
    for s in range(stems):
        eng.set_coeff(s, coeffs(s))
    x = _mag_input(oracle, ntiles, T, F, seed=77 + T + F + ntiles)
    xd = torch.from_numpy(x).cuda()
    masks = eng.forward(xd).cpu().numpy()
    assert np.isfinite(masks).all()
    worst = (0.0, 0.0)
    for s, t in taps:
        worst = max(worst, _check_taps(eng, oracle, coeffs(s), x[t], modes[s], s, t, masks, "T=%d F=%d x%d" % (T, F, ntiles)))
    ks = _layer_kernels(eng, xd)
    if ntiles * stems > 16:
        for name in ("down3", "down4", "down5", "down6"):
            assert ks[name].startswith("srt_enc_wino32<"), (name, ks[name])
        for name in ("up1", "up2", "up3", "up4", "up5"):
            assert ks[name].startswith("srt_dec_wino"), (name, ks[name])
    else:
        assert not any(v.startswith(("srt_dec_wino", "srt_enc_wino")) for v in ks.values()), ks
    eng.close()
    print("geometry %dx%d x%d x%d: worst tap rel-rms %.3g, max-abs/peak %.3g" % (T, F, ntiles, stems, worst[0], worst[1]))


def test_batch_invariant_across_the_up6_and_head_thresholds(oracle, coeffs):
    """under batch_invariant the conv layers are pinned by geometry, but up6 (streamed kernel from 512 column workgroups) and the
    head (rows kernel from 1024 workgroups) still switch with the batch.  Both pairs sum in the same order, so a tile alone and the same tile
    inside a batch of 132 instances (both thresholds crossed, as the kernel names show) must be bit-identical."""
    import torch
    import spleeterrt_amd as srt
    T, F, S, NT = 64, 512, 4, 33
    eng = _engine(F=F, T=T, stem_modes=(1, 0, 1, 0), variant=srt.VARIANT_VST, max_tiles=NT, batch_invariant=True)
    for s in range(S):
        eng.set_coeff(s, coeffs(s))
    x = _mag_input(oracle, NT, T, F, seed=1311)
    xd = torch.from_numpy(x).cuda()
    big = eng.forward(xd).clone()
    kb = _layer_kernels(eng, xd)
    one = xd[17:18].contiguous()
    alone = eng.forward(one).clone()
    ka = _layer_kernels(eng, one)
    assert kb["up6"].startswith("srt_up6_stream_kernel") and not ka["up6"].startswith("srt_up6_stream_kernel"), (kb["up6"], ka["up6"])
    assert kb["up7"].startswith("srt_head_rows_kernel") and not ka["up7"].startswith("srt_head_rows_kernel"), (kb["up7"], ka["up7"])
    assert torch.equal(alone[:, 0], big[:, 17])
    eng.close()


@pytest.mark.parametrize("T,F,ntiles,stems,prec", [
    (64, 1024, 48, 4, "f32"),      # 8 column workgroups per tile x 48 tiles = 384: the smallest batch that takes the streamed form; 8 intervals per column
    (64, 1024, 50, 3, "f32"),      # three stems: the second M tile is half empty (its upper 16 rows are no stem: 8 stores per interval instead of 16)
    (128, 1536, 32, 4, "f32"),     # 12 columns per tile, 16 intervals
    (64, 1024, 48, 4, "f16"),      # fp16 activation storage: conv + bias and act(BN(.)) leave as halves (two 8-byte stores per channel row)
    (64, 1024, 50, 3, "f16"),
])
def test_down1_streamed_form(oracle, coeffs, T, F, ntiles, stems, prec):
    """srt_down1_stream_kernel (down1 walked down 64-pixel output columns, round 4): conv1 of the first, an interior and the last tile of every stem
    against the oracle, the kernel named by the engine, and bit-identity with the tiled kernel a one-tile launch takes (same MFMA chain: the
    switch with the batch size must be invisible, batch_invariant included)."""
    import torch
    import spleeterrt_amd as srt
    modes = tuple((s + 1) % 2 for s in range(stems))
    f16 = prec == "f16"
    eng = _engine(F=F, T=T, stem_modes=modes, variant=srt.VARIANT_VST, max_tiles=ntiles, batch_invariant=not f16, precision=srt.PREC_F16 if f16 else srt.PREC_F32)
    for s in range(stems):
        eng.set_coeff(s, coeffs(s))
    x = _mag_input(oracle, ntiles, T, F, seed=4100 + F + ntiles)
    xd = torch.from_numpy(x).cuda()
    with _env(SPLEETERRT_D1F16=0):                              # (fp16 mode: the fp32-MFMA streamed kernel with halves stored; srt_down1_f16_kernel has its own test)
        _down1_streamed_form_body(oracle, coeffs, eng, x, xd, T, F, ntiles, stems, f16)
    eng.close()


def _down1_streamed_form_body(oracle, coeffs, eng, x, xd, T, F, ntiles, stems, f16):
    import torch
    eng.forward(xd)
    got = {(s, t): eng.tensor("conv1", s, t) for s in range(stems) for t in (0, ntiles // 2 + 1, ntiles - 1)}
    # conv2: in the fp16 mode it pins the act(BN(.)) halves the kernel writes beside conv1 (what down2 reads), and the batch runs down2 with four tiles per
    # workgroup (TPW, csrc/srt_nn3.hip) where the one-tile launch below runs one; in the fp32 mode both run the same kernel: same bits expected either way
    gact = {k: eng.tensor("conv2", *k) for k in got}
    ks = _layer_kernels(eng, xd)
    assert ks["down1"].startswith("srt_down1_stream_kernel<"), ks["down1"]
    if f16:                                                     # down2 of this mode: the four-tiles-per-workgroup form of the one-chunk fp16 layer (csrc/srt_nn3.hip, TPW)
        assert ks["down2"].startswith("srt_enc_c8<32, 8, 1") or ks["down2"].startswith("srt_enc_f16<32, 1, 4, 1, 1, true, 4"), ks["down2"]   # round 6: act1 arrives C8 and down2 runs on the DMA-fed kernel
    lo = oracle.layout()
    for (s, t), g in got.items():
        c = coeffs(s)
        w = c[lo.down[0].w:lo.down[0].w + 25 * 2 * 16]
        ref = oracle.conv5x5_s2(x[t], w, 16) + c[lo.down[0].b:lo.down[0].b + 16][:, None, None]
        assert g.shape == ref.shape
        if f16:                                                 # halves in HBM: one fp16 rounding of an fp32 result
            assert np.abs(g - ref).max() <= 1.0 / 1024 * np.abs(ref).max(), (s, t)
        else:
            assert _rel_rms(g, ref) < TAP_RMS_TOL and np.abs(g - ref).max() < TAP_MAX_TOL * np.abs(ref).max(), (s, t, _rel_rms(g, ref))
    for t in (0, ntiles - 1):                                   # the tiled kernel on the same tile: bit-identical
        one = torch.from_numpy(x[t:t + 1]).cuda()
        eng.forward(one)
        k1 = _layer_kernels(eng, one)
        assert k1["down1"].startswith("srt_enc_mfma2<"), k1["down1"]
        if f16:
            assert k1["down2"].startswith("srt_enc_f16<32, 1, 4, 1, 1, true, 1"), k1["down2"]
        for s in range(stems):
            assert np.array_equal(eng.tensor("conv1", s, 0), got[(s, t)]), (s, t)
            assert np.array_equal(eng.tensor("conv2", s, 0), gact[(s, t)]), (s, t)
