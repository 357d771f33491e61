"""The drop-in C entry points (include/spleeter.h, stftFix.h, Spleeter4Stems.h) beyond single-instance parity:
the public OfflineSTFT tables against the reference's, the failure policy without a GPU (no abort, no CPU path, zeros /
silence with the reference's sample accounting), and two tile-API instances driven concurrently from two host threads
(what the reference's callers do: Executable/main.c:296-330, VST/Source/Spleeter4Stems.c:135)."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

FFT = 4096


class _STFT(C.Structure):                                   # include/stftFix.h == Executable/stftFix.h:19-31
    _fields_ = [("mBitRev", C.c_uint * FFT), ("mPreWindow", C.c_float * FFT), ("mPostWindow", C.c_float * FFT),
                ("mSineTab", C.c_float * FFT), ("threads", C.c_void_p), ("stftThreadData", C.c_void_p),
                ("istftThreadData", C.c_void_p), ("targetCore", C.c_size_t), ("_data", C.c_void_p * 2),
                ("shared_info", C.c_void_p)]


def _lib():
    import spleeterrt_amd
    return spleeterrt_amd.load_library()


def _has_gpu():
    import torch
    return torch.cuda.is_available()


def test_offline_stft_tables_match_reference(oracle):
    """a10: InitSTFT fills the PUBLIC tables (mBitRev, mPreWindow, mPostWindow, mSineTab) exactly as stftFix.c:302-313 does —
    bit for bit against the CPU oracle's tables and, where the reference build exists, against the reference's own InitSTFT.
    (Without a GPU InitSTFT reports the missing device on stderr and leaves `threads` NULL; the tables are filled first.)"""
    L = _lib()
    st = _STFT()
    L.InitSTFT.argtypes = [C.POINTER(_STFT), C.c_size_t]
    L.FreeSTFT.argtypes = [C.POINTER(_STFT)]
    L.InitSTFT(C.byref(st), 3)
    assert st.targetCore == 3
    mine = {k: np.array(getattr(st, k)) for k in ("mBitRev", "mPreWindow", "mPostWindow", "mSineTab")}
    L.FreeSTFT(C.byref(st))
    t = oracle.tables()
    for k, o in (("mBitRev", t.rev), ("mPreWindow", t.pre), ("mPostWindow", t.post), ("mSineTab", t.sine)):
        assert np.array_equal(mine[k], np.array(o)), k
    if oracle.ref_path("exe"):
        R = C.CDLL(oracle.ref_path("exe"))
        rs = _STFT()
        R.InitSTFT.argtypes = [C.POINTER(_STFT), C.c_size_t]
        R.InitSTFT(C.byref(rs), 1)
        for k in mine:
            assert np.array_equal(mine[k], np.array(getattr(rs, k))), "vs reference: " + k
        R.FreeSTFT.argtypes = [C.POINTER(_STFT)]
        R.FreeSTFT(C.byref(rs))


def test_failure_policy_without_gpu(oracle, coeffs, capfd):
    """No GPU: nothing aborts and nothing is computed on the CPU — the tile API returns a zero mask, stft/istft return zeroed
    planes of the right size, the streaming surface emits silence with exact sample accounting, and srtLastError() says why."""
    if _has_gpu():
        pytest.skip("GPU present")
    L = _lib()
    L.srtLastError.restype = C.c_char_p
    T, F = 64, 512
    # tile API
    L.allocateSpleeterStr.restype = C.c_void_p
    nn = C.c_void_p(L.allocateSpleeterStr())
    c = np.ascontiguousarray(coeffs(0))
    L.initSpleeter.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
    L.initSpleeter(nn, F, T, 1, c.ctypes.data)
    assert b"no HIP device" in L.srtLastError()
    x = np.ones((2, T, F), np.float32)
    y = np.full((2, T, F), 7.0, np.float32)
    L.processSpleeter.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.processSpleeter(nn, x.ctypes.data, y.ctypes.data)
    assert np.all(y == 0)
    mask = C.POINTER(C.c_float)()
    L.getMaskPtr.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_float))]
    L.getMaskPtr(nn, C.byref(mask))
    assert bool(mask)
    L.freeSpleeter.argtypes = [C.c_void_p]
    L.freeSpleeter(nn)
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(nn)
    # STFT API
    st = _STFT()
    L.InitSTFT.argtypes = [C.POINTER(_STFT), C.c_size_t]
    L.InitSTFT(C.byref(st), 1)
    n = 4096 * 3
    a = np.ones(n, np.float32)
    ptrs = [C.POINTER(C.c_float)() for _ in range(4)]
    L.stft.restype = C.c_size_t
    L.stft.argtypes = [C.POINTER(_STFT), C.c_void_p, C.c_void_p, C.c_size_t] + [C.POINTER(C.POINTER(C.c_float))] * 4
    rows = L.stft(C.byref(st), a.ctypes.data, a.ctypes.data, n, *[C.byref(p) for p in ptrs])
    assert rows == 12
    for p in ptrs:
        assert np.all(np.ctypeslib.as_array(p, shape=(rows, FFT)) == 0)
        libc.free(p)
    L.FreeSTFT.argtypes = [C.POINTER(_STFT)]
    L.FreeSTFT(C.byref(st))
    # streaming surface: 3 calls of 700 samples -> 700 + 700 + 700 in, segments appear after each full 1024-sample hop
    msr = C.create_string_buffer(4096)
    cs = [np.ascontiguousarray(coeffs(0))] * 4
    prov = (C.c_void_p * 4)(*[q.ctypes.data for q in cs])
    L.Spleeter4StemsInit.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.Spleeter4StemsProcessSamples.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.Spleeter4StemsFree.argtypes = [C.c_void_p]
    L.Spleeter4StemsInit(msr, F, T, prov)
    out = np.full((8, 2100), 5.0, np.float32)
    pos = 0
    for k in range(3):
        ptr = (C.c_void_p * 8)(*[out[j].ctypes.data + 4 * pos for j in range(8)])
        L.Spleeter4StemsProcessSamples(msr, a.ctypes.data, a.ctypes.data, 700, ptr)
        pos += 700
    L.Spleeter4StemsFree(msr)
    # the reference writes nothing before the first finished hop (first call: 700 < 1024), then min(queued, asked) samples per call
    assert np.all(out[:, :700] == 5.0)
    assert np.all(out[:, 700:2100] == 0.0)
    err = capfd.readouterr().err
    assert "no CPU fallback" in err


@pytest.mark.gpu
def test_two_tile_api_instances_from_two_threads(oracle, coeffs):
    """Two _spleeter instances with different weights and activation modes, each driven by its own host thread for several
    rounds at the same time; every result must equal the CPU oracle's (and therefore never mix the two instances' state)."""
    L = _lib()
    os.environ["SPLEETERRT_VARIANT"] = "vst"
    T, F = 64, 512
    L.allocateSpleeterStr.restype = C.c_void_p
    L.initSpleeter.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
    L.processSpleeter.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.freeSpleeter.argtypes = [C.c_void_p]
    modes = (1, 0)
    cs = [np.ascontiguousarray(coeffs(k)) for k in range(2)]
    xs = [np.abs(oracle.lcg(100 + k, 2 * T * F, 6.0)).reshape(2, T, F).astype(np.float32) for k in range(2)]
    refs = [oracle.forward(cs[k], xs[k], modes[k], oracle.VARIANT_VST) for k in range(2)]
    nns = [C.c_void_p(L.allocateSpleeterStr()) for _ in range(2)]
    errs = [[], []]
    start = threading.Barrier(2)

    def work(k):
        L.initSpleeter(nns[k], F, T, modes[k], cs[k].ctypes.data)        # concurrent init as well (main.c:554-573)
        y = np.empty((2, T, F), np.float32)
        start.wait()
        for _ in range(25):
            y.fill(-1.0)
            L.processSpleeter(nns[k], xs[k].ctypes.data, y.ctypes.data)
            errs[k].append(float(np.abs(y - refs[k]).max()))
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    for k in range(2):
        L.freeSpleeter(nns[k])
        libc.free(nns[k])
    os.environ.pop("SPLEETERRT_VARIANT")
    for k in range(2):
        assert len(errs[k]) == 25 and max(errs[k]) <= 2e-4, "instance %d: %r" % (k, max(errs[k]) if errs[k] else None)
