"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, when present, the real reference
build (oracle/_ref/*.so, compiled from /root/reference by oracle/Makefile).

TEST INFRASTRUCTURE ONLY: importers are tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (spleeterrt_amd/) must never import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
COEFF_FLOATS = 9822725
FFT, HOP, HALF = 4096, 1024, 2049
VARIANT_EXE, VARIANT_VST = 0, 1

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u16p = np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when the reference sources are present)."""
    so = os.path.join(HERE, "liboracle.so")
    src = [os.path.join(HERE, f) for f in ("spleeter_oracle.c", "spleeter_oracle.h")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src)
    if stale or (os.path.isdir("/root/reference/Executable") and not os.path.exists(os.path.join(HERE, "_ref", "libspleeter_ref.so"))):
        subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    return so


class _Taps(C.Structure):
    _fields_ = [("conv", C.c_void_p * 6), ("act", C.c_void_p * 5), ("up", C.c_void_p * 6)]


class _LayerOff(C.Structure):
    _fields_ = [("w", C.c_size_t), ("b", C.c_size_t), ("bn", C.c_size_t), ("cin", C.c_int), ("cout", C.c_int)]


class _Layout(C.Structure):
    _fields_ = [("down", _LayerOff * 6), ("up", _LayerOff * 6), ("head_w", C.c_size_t), ("head_b", C.c_size_t)]


class _Tables(C.Structure):
    _fields_ = [("rev", C.c_uint * FFT), ("pre", C.c_float * FFT), ("post", C.c_float * FFT), ("sine", C.c_float * FFT)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_lcg_fill.restype = C.c_uint32
        L.orc_lcg_fill.argtypes = [C.c_uint32, _f32p, C.c_size_t, C.c_float]
        L.orc_synth_coeff_fp16.argtypes = [_u16p, C.c_int]
        L.orc_fp16_expand.argtypes = [_u16p, _f32p, C.c_size_t]
        L.orc_synth_coeff.argtypes = [_f32p, C.c_int]
        L.orc_synth_audio.argtypes = [_f32p, _f32p, C.c_size_t, C.c_uint32, C.c_int]
        L.orc_get_layout.argtypes = [C.POINTER(_Layout)]
        L.orc_conv5x5_s2.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, _f32p]
        L.orc_tconv5x5_s2.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, _f32p]
        L.orc_conv4x4_d2.argtypes = [_f32p, C.c_int, C.c_int, _f32p, _f32p]
        for fn in (L.orc_sigmoid_lut, L.orc_sigmoid_exact):
            fn.restype = C.c_float
            fn.argtypes = [C.c_float]
        L.orc_act.restype = C.c_float
        L.orc_act.argtypes = [C.c_float, C.c_int, C.c_int]
        L.orc_forward.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p, C.POINTER(_Taps)]
        L.orc_stft_init.argtypes = [C.POINTER(_Tables)]
        L.orc_fht4096.argtypes = [_f32p, _f32p]
        L.orc_stft_frames.restype = C.c_size_t
        L.orc_stft_frames.argtypes = [C.c_size_t]
        L.orc_stft.restype = C.c_size_t
        L.orc_stft.argtypes = [C.POINTER(_Tables), _f32p, _f32p, C.c_size_t, _f32p, _f32p, _f32p, _f32p]
        L.orc_istft.restype = C.c_size_t
        L.orc_istft.argtypes = [C.POINTER(_Tables), _f32p, _f32p, _f32p, _f32p, C.c_size_t, _f32p, _f32p]
        L.orc_magnitude_tile.argtypes = [_f32p, _f32p, _f32p, _f32p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, _f32p]
        L.orc_mask_apply_tile.argtypes = [_f32p, _f32p, _f32p, _f32p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, _f32p, C.c_float]
        L.orc_process_spectrogram.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, _f32p, _f32p, _f32p, _f32p, C.c_float]
        _lib = L
    return _lib


# ---------------------------------------------------------------- synthetic data
def lcg(seed, n, scale=1.0):
    out = np.empty(n, np.float32)
    lib().orc_lcg_fill(seed, out, n, scale)
    return out


def synth_coeff_fp16(stem):
    h = np.empty(COEFF_FLOATS, np.uint16)
    lib().orc_synth_coeff_fp16(h, stem)
    return h


def fp16_expand(h):
    out = np.empty(h.size, np.float32)
    lib().orc_fp16_expand(np.ascontiguousarray(h, np.uint16), out, h.size)
    return out


def synth_coeff(stem):
    out = np.empty(COEFF_FLOATS, np.float32)
    lib().orc_synth_coeff(out, stem)
    return out


def synth_audio(n, seed=777, tones=False):
    L = np.empty(n, np.float32)
    R = np.empty(n, np.float32)
    lib().orc_synth_audio(L, R, n, seed, int(tones))
    return L, R


def layout():
    lo = _Layout()
    lib().orc_get_layout(C.byref(lo))
    return lo


# ---------------------------------------------------------------- network
def conv5x5_s2(x, w, cout):
    cin, H, W = x.shape
    y = np.empty((cout, H // 2, W // 2), np.float32)
    lib().orc_conv5x5_s2(np.ascontiguousarray(x), cin, H, W, np.ascontiguousarray(w), cout, y)
    return y


def tconv5x5_s2(x, w, cout):
    cin, H, W = x.shape
    y = np.empty((cout, 2 * H, 2 * W), np.float32)
    lib().orc_tconv5x5_s2(np.ascontiguousarray(x), cin, H, W, np.ascontiguousarray(w), cout, y)
    return y


def conv4x4_d2(x, w):
    H, W = x.shape
    y = np.empty((2, H, W), np.float32)
    lib().orc_conv4x4_d2(np.ascontiguousarray(x), H, W, np.ascontiguousarray(w), y)
    return y


def forward(coeff, x, stem_mode, variant=VARIANT_EXE, want_taps=False):
    """x: [2][T][F] magnitudes -> mask [2][T][F] (Executable/spleeter.c:177-301)."""
    _, T, F = x.shape
    y = np.empty_like(x)
    taps = None
    keep = {}
    if want_taps:
        taps = _Taps()
        lo = layout()
        for i in range(6):
            co = lo.down[i].cout
            a = np.empty((co, T >> (i + 1), F >> (i + 1)), np.float32)
            keep["conv%d" % (i + 1)] = a
            taps.conv[i] = a.ctypes.data
            if i < 5:
                b = np.empty_like(a)
                keep["act%d" % (i + 1)] = b
                taps.act[i] = b.ctypes.data
        for i in range(6):
            co = lo.up[i].cout
            a = np.empty((co, T >> (5 - i), F >> (5 - i)), np.float32)
            keep["up%d" % (i + 1)] = a
            taps.up[i] = a.ctypes.data
    lib().orc_forward(np.ascontiguousarray(coeff), F, T, stem_mode, variant, np.ascontiguousarray(x), y,
                      C.byref(taps) if taps is not None else None)
    return (y, keep) if want_taps else y


# ---------------------------------------------------------------- DSP
_tables = None


def tables():
    global _tables
    if _tables is None:
        _tables = _Tables()
        lib().orc_stft_init(C.byref(_tables))
    return _tables


def stft(L, R):
    """-> (re, im) each [2][rows][4096] (bins > 2048 zero), rows = ceil(n/1024). stftFix.c:363-495"""
    n = L.size
    rows = lib().orc_stft_frames(n)
    re = np.zeros((2, rows, FFT), np.float32)
    im = np.zeros((2, rows, FFT), np.float32)
    lib().orc_stft(C.byref(tables()), np.ascontiguousarray(L), np.ascontiguousarray(R), n, re[0], im[0], re[1], im[1])
    return re, im


def istft(re, im):
    frames = re.shape[1]
    n = frames * HOP + FFT - HOP
    out = np.zeros((2, n), np.float32)
    lib().orc_istft(C.byref(tables()), re[0], im[0], re[1], im[1], frames, out[0], out[1])
    return out


def magnitude_tile(re, im, row0, T, F):
    mag = np.empty((2, T, F), np.float32)
    lib().orc_magnitude_tile(re[0], im[0], re[1], im[1], re.shape[1], row0, T, F, mag)
    return mag


def process_spectrogram(coeff, re, im, F, T, stem_mode, variant=VARIANT_EXE, unaffected=0.1):
    """In-place processMT restatement (main.c:444-541)."""
    lib().orc_process_spectrogram(np.ascontiguousarray(coeff), F, T, stem_mode, variant, re.shape[1], re[0], im[0], re[1], im[1], unaffected)


def cli_separate(coeff_net0, coeff_net1, L, R, F, T, stems, variant=VARIANT_EXE, unaffected=0.1):
    """The offline CLI's two- and three-output flows restated over the oracle's own stft / processMT / istft
    (Executable/main.c:776-798 and :845-928).  net0 = coeffProvPtr2 (drum, mode 1), net1 = coeffProvPtr1 (vocal, mode 0)
    (main.c:759-760).  Returns [stems][2][rows*1024+3072] in file order: (Vocal, Accompaniment) or (Drum, Vocal, Accompaniment).
    The reference's main() cannot be built here (its embedded model.c is absent), so this flow is pinned by reading only;
    every stage it calls is pinned on its own against oracle/_ref."""
    L = np.ascontiguousarray(L, np.float32)
    R = np.ascontiguousarray(R, np.float32)
    n = L.size
    re, im = stft(L, R)
    if stems == 2:
        process_spectrogram(coeff_net1, re, im, F, T, 0, variant, unaffected)                    # main.c:779
        vocal = istft(re, im)
        acc = -vocal.copy()
        acc[0, :n] = L - vocal[0, :n]                                                            # main.c:794-798
        acc[1, :n] = R - vocal[1, :n]
        return np.stack([vocal, acc])
    ore, oim = re.copy(), im.copy()                                                              # main.c:849-856
    process_spectrogram(coeff_net0, re, im, F, T, 1, variant, unaffected)                        # main.c:858
    ore -= re                                                                                    # main.c:860-866
    oim -= im
    drum = istft(re, im)
    accvocal = istft(ore, oim)                                                                   # main.c:881
    process_spectrogram(coeff_net1, ore, oim, F, T, 0, variant, unaffected)                      # main.c:911
    vocal = istft(ore, oim)
    return np.stack([drum, vocal, accvocal - vocal])                                             # main.c:924-928


def ratio_mask(masks, eps=1e-10):
    """Official-Spleeter cross-stem normalisation (absent from the reference, README.MD:82-85): masks [S, ...]."""
    sq = masks.astype(np.float32) ** 2
    return ((sq + np.float32(eps / masks.shape[0])) / (sq.sum(axis=0, dtype=np.float32) + np.float32(eps))).astype(np.float32)


# ---------------------------------------------------------------- real reference (oracle/_ref)
def ref_path(flavour="exe"):
    name = {"exe": "libspleeter_ref.so", "avx2": "libspleeter_ref_avx2.so", "vst": "libspleeter_ref_vst.so",
            "stream": "libspleeter_ref_stream.so"}[flavour]
    p = os.path.join(HERE, "_ref", name)
    return p if os.path.exists(p) else None


class RefNet:
    """The reference's own tile API (Executable/spleeter.h:63-69) from oracle/_ref."""

    def __init__(self, coeff, F, T, stem_mode, flavour="exe"):
        p = ref_path(flavour)
        if p is None:
            raise FileNotFoundError("oracle/_ref not built")
        self.L = C.CDLL(p)
        self.L.allocateSpleeterStr.restype = C.c_void_p
        self.L.getCoeffSize.restype = C.c_size_t
        assert self.L.getCoeffSize() == COEFF_FLOATS * 4
        self.coeff = np.ascontiguousarray(coeff, np.float32)      # borrowed by the instance: keep alive
        self.nn = C.c_void_p(self.L.allocateSpleeterStr())
        if flavour in ("vst", "stream"):
            self.L.initSpleeter.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        else:
            self.L.initSpleeter.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
        self.L.initSpleeter(self.nn, F, T, stem_mode, self.coeff.ctypes.data)
        self.L.processSpleeter.argtypes = [C.c_void_p, _f32p, _f32p]
        self.F, self.T = F, T

    def __call__(self, x):
        y = np.empty((2, self.T, self.F), np.float32)
        self.L.processSpleeter(self.nn, np.ascontiguousarray(x, np.float32), y)
        return y

    def close(self):
        if self.nn:
            self.L.freeSpleeter.argtypes = [C.c_void_p]
            self.L.freeSpleeter(self.nn)
            C.CDLL(None).free.argtypes = [C.c_void_p]
            C.CDLL(None).free(self.nn)
            self.nn = None


class _RefSTFTStruct(C.Structure):
    # Executable/stftFix.h:19-31
    _fields_ = [("mBitRev", C.c_uint * FFT), ("mPreWindow", C.c_float * FFT), ("mPostWindow", C.c_float * FFT),
                ("mSineTab", C.c_float * FFT), ("threads", C.c_void_p), ("stftThreadData", C.c_void_p),
                ("istftThreadData", C.c_void_p), ("targetCore", C.c_size_t), ("_data", C.c_void_p * 2),
                ("shared_info", C.c_void_p)]


class RefSTFT:
    def __init__(self, cores=1, flavour="exe"):
        p = ref_path(flavour)
        if p is None:
            raise FileNotFoundError("oracle/_ref not built")
        self.L = C.CDLL(p)
        self.st = _RefSTFTStruct()
        self.L.InitSTFT.argtypes = [C.POINTER(_RefSTFTStruct), C.c_size_t]
        self.L.InitSTFT(C.byref(self.st), cores)
        self.L.stft.restype = C.c_size_t
        self.L.stft.argtypes = [C.POINTER(_RefSTFTStruct), _f32p, _f32p, C.c_size_t] + [C.POINTER(C.POINTER(C.c_float))] * 4
        self.L.istft.restype = C.c_size_t
        self.L.istft.argtypes = [C.POINTER(_RefSTFTStruct), _f32p, _f32p, _f32p, _f32p, C.c_size_t] + [C.POINTER(C.POINTER(C.c_float))] * 2
        self.libc = C.CDLL(None)
        self.libc.free.argtypes = [C.c_void_p]

    def stft(self, L, R):
        ptrs = [C.POINTER(C.c_float)() for _ in range(4)]
        rows = self.L.stft(C.byref(self.st), np.ascontiguousarray(L), np.ascontiguousarray(R), L.size, *[C.byref(p) for p in ptrs])
        planes = [np.ctypeslib.as_array(p, shape=(rows, FFT)).copy() for p in ptrs]
        for p in ptrs:
            self.libc.free(p)
        re = np.stack([planes[0], planes[2]])
        im = np.stack([planes[1], planes[3]])
        return re, im

    def istft(self, re, im):
        re = re.copy(); im = im.copy()                      # the reference clobbers its inputs
        frames = re.shape[1]
        ptrs = [C.POINTER(C.c_float)() for _ in range(2)]
        n = self.L.istft(C.byref(self.st), re[0], im[0], re[1], im[1], frames, C.byref(ptrs[0]), C.byref(ptrs[1]))
        out = np.stack([np.ctypeslib.as_array(p, shape=(n,)).copy() for p in ptrs])
        for p in ptrs:
            self.libc.free(p)
        return out

    def close(self):
        self.L.FreeSTFT.argtypes = [C.POINTER(_RefSTFTStruct)]
        self.L.FreeSTFT(C.byref(self.st))
