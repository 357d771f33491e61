"""CPU tests (no GPU): the oracle is pinned (a) against the committed golden vectors generated from the real
reference build, and (b) against that build itself when oracle/_ref is present (build container and GPU box)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz")


def _stats(a):
    a64 = a.astype(np.float64).ravel()
    w = np.cos(np.arange(a64.size) * 0.61803398875)
    return np.array([a64.sum(), (a64 ** 2).sum(), (a64 * w).sum(), a64.min(), a64.max()])


def _mag(oracle, ntiles, T, F, seed):
    x = np.abs(oracle.lcg(seed, ntiles * 2 * T * F, 6.0)).reshape(ntiles, 2, T, F)
    x[:, :, ::7, ::13] *= 8.0
    return np.ascontiguousarray(x, np.float32)


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_layout_and_sizes(oracle):
    lo = oracle.layout()
    assert lo.head_b + 2 == oracle.COEFF_FLOATS == 9822725           # sizeof(spleeterCoeff) = 39 290 900 B
    assert lo.down[5].bn == lo.up[0].w                               # down6 has no batchNorm
    assert lo.up[0].cin == 512 and lo.up[0].cout == 256


def test_lcg_known_values(oracle):
    # s <- s*1664525 + 1013904223; u = (s>>8)/2^24 - 0.5  
    s, exp = 777, []
    for _ in range(5):
        s = (s * 1664525 + 1013904223) & 0xffffffff
        exp.append((s >> 8) / 16777216.0 - 0.5)
    assert np.allclose(oracle.lcg(777, 5), np.array(exp, np.float32), atol=0)


def test_fp16_expand_semantics(oracle):
    h = np.array([0x0000, 0x8000, 0x3c00, 0xbc00, 0x0001, 0x83ff, 0x0400, 0x7bff, 0x3555], np.uint16)
    got = oracle.fp16_expand(h)
    exp = h.view(np.float16).astype(np.float32)
    exp[4] = 0.0; exp[5] = -0.0                                      # half-denormals flush to +-0 (main.c:431)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    # Inf/NaN are NOT special-cased by the reference: exponent 31 just re-biases (main.c:428)
    assert oracle.fp16_expand(np.array([0x7c00], np.uint16))[0] == 65536.0


@pytest.mark.parametrize("variant", ["exe", "vst"])
@pytest.mark.parametrize("stem,mode", [(0, 0), (1, 1)])
def test_forward_matches_golden(oracle, coeffs, gold, variant, stem, mode):
    x = _mag(oracle, 1, 64, 512, 4242)[0]
    y = oracle.forward(coeffs(stem), x, mode, oracle.VARIANT_EXE if variant == "exe" else oracle.VARIANT_VST)
    key = "fwd_%s_stem%d_mode%d" % (variant, stem, mode)
    ref = gold[key + "_sub"]
    tol = 0.0 if variant == "vst" else 2.5e-7                          # exact-sigmoid flavour is bit-exact; LUT regenerated (<= 6e-8)
    assert np.abs(y.ravel()[::7] - ref).max() <= tol
    assert np.allclose(_stats(y), gold[key + "_stats"], rtol=1e-6, atol=1e-6)
    assert 0.2 < y.std() < 0.5                                       # masks are not vacuous (SURVEY §7 hard part iv)


def test_activations_match_golden(oracle, gold):
    xs = gold["act_x"]
    sig = np.array([oracle.lib().orc_sigmoid_lut(float(v)) for v in xs], np.float32)
    assert np.abs(sig - gold["act_fastSigmoid"]).max() <= 1e-7
    for name, kind in (("leakyReLU", 0), ("ReLU", 1), ("ELU", 2)):
        got = np.array([oracle.lib().orc_act(float(v), kind, 0) for v in xs], np.float32)
        assert np.array_equal(got, gold["act_" + name]), name
    # VST flavour: exact logistic, no ELU clamp
    assert abs(oracle.lib().orc_sigmoid_exact(0.3) - 1 / (1 + np.exp(-0.3))) < 1e-7
    assert oracle.lib().orc_act(-15.5, 2, 1) != -1.0 and oracle.lib().orc_act(-15.5, 2, 0) == -1.0


def test_stft_istft_match_golden(oracle, gold):
    n = int(gold["stft_n"][0])
    L, R = oracle.synth_audio(n, 777, True)
    re, im = oracle.stft(L, R)
    peak = np.abs(gold["stft_re_sub"]).max()
    assert np.abs(re[:, :, :2049].ravel()[::5] - gold["stft_re_sub"]).max() <= 1e-6 * peak
    assert np.abs(im[:, :, :2049].ravel()[::5] - gold["stft_im_sub"]).max() <= 1e-6 * peak
    assert np.all(re[:, :, 2049:] == 0) and np.all(im[:, :, 2049:] == 0)
    y = oracle.istft(re, im)
    assert np.abs(y.ravel()[::3] - gold["istft_sub"]).max() <= 1e-6 * np.abs(gold["istft_sub"]).max()
    # convention check against numpy: re = Re(rfft(x*hann))/4096, im = -Im(...)/4096
    w = 0.5 * (1 - np.cos(2 * np.pi * (np.arange(4096) + 0.5) / 4096))
    fr = np.fft.rfft(L[2048:2048 + 4096].astype(np.float64) * w) / 4096
    assert np.abs(re[0, 2, :2049] - fr.real).max() < 1e-7 and np.abs(im[0, 2, :2049] + fr.imag).max() < 1e-7


def test_fht_matches_golden(oracle, gold):
    a = oracle.lcg(99, 4096, 2.0)
    t = oracle.tables()
    br = np.empty(4096, np.float32)
    br[np.array(t.rev[:], np.int64)] = a
    oracle.lib().orc_fht4096(br, np.array(t.sine[:], np.float32))
    assert np.abs(br - gold["fht_out"]).max() <= 2e-6 * np.abs(gold["fht_out"]).max()
    # Hartley definition: H[k] = sum a[n] cas(2 pi n k / N)
    k = np.array([0, 1, 5, 2047, 2048, 4095])
    ang = 2 * np.pi * np.outer(k, np.arange(4096)) / 4096
    assert np.abs(br[k] - ((np.cos(ang) + np.sin(ang)) @ a.astype(np.float64))).max() < 1e-3


def test_tail_tile_and_mask_rules(oracle, coeffs):
    """processMT semantics: rows past the end are zero magnitudes and are never written back (main.c:496-537);
    bins >= F are scaled by unaffectedWeight (main.c:486-493)."""
    T, F = 64, 512
    n = 4096 * 20 + 8192                                             # 88 rows: one full tile + 24-row tail
    L, R = oracle.synth_audio(n, 5, False)
    re, im = oracle.stft(L, R)
    r0, i0 = re.copy(), im.copy()
    oracle.process_spectrogram(coeffs(1), re, im, F, T, 1, oracle.VARIANT_VST, 0.1)
    assert np.allclose(re[:, :, F:2049], r0[:, :, F:2049] * np.float32(0.1), rtol=0, atol=0)
    mag = oracle.magnitude_tile(r0, i0, 64, T, F)
    assert np.all(mag[:, 24:, :] == 0) and np.any(mag[:, :24, :] != 0)
    ratio = np.abs(re[:, :, :F]) / (np.abs(r0[:, :, :F]) + 1e-30)
    assert ratio.max() <= 1.0 + 1e-6                                 # masks live in [0,1]


# ------------------------------------------------------------------ against the real reference build, when present
def _need_ref(oracle, flavour="exe"):
    if oracle.ref_path(flavour) is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")


def test_oracle_vs_reference_forward(oracle, coeffs):
    _need_ref(oracle)
    x = _mag(oracle, 1, 64, 512, 31)[0]
    for flavour, variant, tol in (("vst", oracle.VARIANT_VST, 0.0), ("exe", oracle.VARIANT_EXE, 2.5e-7)):
        for stem, mode in ((2, 0), (3, 1)):
            net = oracle.RefNet(coeffs(stem), 512, 64, mode, flavour)
            yr = net(x)
            net.close()
            assert np.abs(oracle.forward(coeffs(stem), x, mode, variant) - yr).max() <= tol


@pytest.mark.parametrize("T,F", [(256, 1024), (512, 1024), (256, 1536), (256, 2048)])
def test_oracle_vs_reference_forward_at_shipped_tile_sizes(oracle, coeffs, T, F):
    """The restatement pinned against the real reference network at the tile sizes the GPU tests check it at: BASELINE's 256 x 1024, the CLI's
    default 512 x 1024 (main.c:701-702), the plugin's 256 x 1536 (PluginProcessor.cpp:124) and the CLI's widest, 2048 bins (main.c:745-748)."""
    _need_ref(oracle)
    x = _mag(oracle, 1, T, F, 31 + T + F)[0]
    for flavour, variant, tol, stem, mode in (("vst", oracle.VARIANT_VST, 0.0, 3, 1), ("exe", oracle.VARIANT_EXE, 2.5e-7, 2, 0)):
        net = oracle.RefNet(coeffs(stem), F, T, mode, flavour)
        yr = net(x)
        net.close()
        assert np.abs(oracle.forward(coeffs(stem), x, mode, variant) - yr).max() <= tol


def test_oracle_vs_reference_stft(oracle):
    _need_ref(oracle)
    n = 4096 * 4 + 8192
    L, R = oracle.synth_audio(n, 12345, True)
    st = oracle.RefSTFT(1)
    rre, rim = st.stft(L, R)
    re, im = oracle.stft(L, R)
    peak = np.abs(rre).max()
    assert np.abs(re - rre).max() <= 1e-6 * peak and np.abs(im - rim).max() <= 1e-6 * peak
    ro = st.istft(rre, rim)
    assert np.abs(oracle.istft(re, im) - ro).max() <= 1e-6 * np.abs(ro).max()
    # property the reference implies but never checks: multi-threaded stft == single-threaded, bitwise
    st3 = oracle.RefSTFT(3)
    r3, i3 = st3.stft(L, R)
    assert np.array_equal(r3, rre) and np.array_equal(i3, rim)
    st.close(); st3.close()


def test_ratio_mask_restatement(oracle):
    """(m_s^2 + eps/S) / (sum m_j^2 + eps): sums to one over stems, equal masks -> 1/S, zeros stay finite."""
    S = 4
    m = (oracle.lcg(5, S * 1000, 1.0) + 0.5).reshape(S, 1000).astype(np.float32)
    r = oracle.ratio_mask(m)
    assert r.dtype == np.float32 and np.abs(r.sum(axis=0) - 1.0).max() < 1e-6
    assert np.allclose(oracle.ratio_mask(np.full((S, 8), 0.3, np.float32)), 1.0 / S, atol=1e-7)
    assert np.allclose(oracle.ratio_mask(np.zeros((S, 8), np.float32)), 1.0 / S, atol=1e-7)
    a = np.zeros((2, 4), np.float32)
    a[0] = 0.9
    assert np.allclose(oracle.ratio_mask(a)[0], 1.0, atol=1e-6)


@pytest.mark.parametrize("stems", [2, 3])
def test_cli_flow_restatement_vs_reference_linked_harness(oracle, tmp_path, stems):
    """oracle.cli_separate (main.c:776-798, 845-928 restated) against host/offline_main.c linked to the REAL reference
    (oracle/_ref): same padded input, same fp16-container weights, EXE flavour.  Pins the residual arithmetic and the
    output order of the two- and three-output flows."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = os.path.join(root, "host", "offline_ref")
    if oracle.ref_path("exe") is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    if not os.path.exists(ref):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "host"), "offline_ref"])
    T, F, n = 64, 512, 44100 + 321
    h0, h1 = oracle.synth_coeff_fp16(1), oracle.synth_coeff_fp16(0)          # container order: net[0] drum, net[1] vocal
    np.concatenate([h0, h1]).tofile(tmp_path / "w.f16")
    L, R = oracle.synth_audio(n, 321, True)
    np.stack([L, R], 1).astype(np.float32).tofile(tmp_path / "in.f32")
    subprocess.check_call([ref, str(T), str(F), str(stems), str(tmp_path / "w.f16"), str(tmp_path / "in.f32"), str(tmp_path / "o")])
    final = 4096 * ((n + 4095) // 4096) + 8192
    pL, pR = np.zeros(final, np.float32), np.zeros(final, np.float32)
    pL[4096:4096 + n], pR[4096:4096 + n] = L, R
    got = oracle.cli_separate(oracle.fp16_expand(h0), oracle.fp16_expand(h1), pL, pR, F, T, stems, oracle.VARIANT_EXE, 0.1)
    names = ["Vocal", "Accompaniment"] if stems == 2 else ["Drum", "Vocal", "Accompaniment"]
    for k, nm in enumerate(names):
        r = np.fromfile(tmp_path / ("o_%s.f32" % nm), np.float32).reshape(-1, 2)
        a = got[k][:, 4096:4096 + n].T
        assert a.shape == r.shape
        assert np.abs(a - r).max() <= 2e-6 * max(np.abs(r).max(), 1e-3), "%s: %g" % (nm, np.abs(a - r).max())


def test_oracle_fp16_expand_every_pattern(oracle):
    """The oracle's restatement of f32Decompress (Executable/main.c:423-434) against a table worked out by hand for all 65 536 half patterns:
    exponent 0 -> signed zero, exponent 1..30 -> the IEEE value, exponent 31 re-biased like a normal (+-65536 * (1 + m/1024); no Inf/NaN)."""
    h = np.arange(65536, dtype=np.uint32)
    sign, ex, man = h >> 15, (h >> 10) & 31, h & 1023
    bits = np.where(ex == 0, sign << 31, (sign << 31) | ((ex + 112) << 23) | (man << 13)).astype(np.uint32)     # 112 = 127 - 15: main.c:430 adds 0x38000000
    got = oracle.fp16_expand(h.astype(np.uint16))
    assert np.array_equal(got.view(np.uint32), bits)
    with np.errstate(all="ignore"):
        ieee = h.astype(np.uint16).view(np.float16).astype(np.float32)
    normal = (ex >= 1) & (ex <= 30)
    assert np.array_equal(got[normal], ieee[normal])                                     # and numpy's own conversion agrees wherever IEEE defines a finite normal
    assert np.all(got[ex == 0] == 0) and np.all(np.signbit(got[(ex == 0) & (sign == 1)]))
    assert np.all(np.isfinite(got)) and np.abs(got[ex == 31]).min() == 65536.0
