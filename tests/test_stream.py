"""GPU parity of the real-time streaming surface (include/Spleeter4Stems.h) against the real reference streaming
engine (oracle/_ref/libspleeter_ref_stream.so = VST/Source/Spleeter4Stems.c + VST network, CPU_GEMM=1)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(lib, struct_bytes, coeffs, F, T, L, R, chunks):
    msr = C.create_string_buffer(struct_bytes)
    prov = (C.c_void_p * 4)(*[c.ctypes.data for c in coeffs])
    lib.Spleeter4StemsInit.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.Spleeter4StemsProcessSamples.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.Spleeter4StemsFree.argtypes = [C.c_void_p]
    lib.Spleeter4StemsInit(msr, F, T, prov)
    n = L.size
    out = np.zeros((8, n), np.float32)
    pos = 0
    i = 0
    while pos < n:
        c = min(chunks[i % len(chunks)], n - pos)
        i += 1
        ptrs = (C.c_void_p * 8)(*[out[j].ctypes.data + 4 * pos for j in range(8)])
        lib.Spleeter4StemsProcessSamples(msr, L.ctypes.data + 4 * pos, R.ctypes.data + 4 * pos, c, ptrs)
        pos += c
    lib.Spleeter4StemsFree(msr)
    return out


@pytest.mark.parametrize("T,F,chunks", [(64, 512, (1024,)), (64, 512, (300, 724, 1024, 512, 17)),
                                        (256, 1536, (480, 1024, 544))])   # last: the geometry the plugin ships (PluginProcessor.cpp:124)
def test_streaming_matches_reference(oracle, coeffs, T, F, chunks):
    if oracle.ref_path("stream") is None:
        pytest.skip("oracle/_ref/libspleeter_ref_stream.so not built")
    import spleeterrt_amd
    hops = 3 * T + 9      # the first 3T hops (masks of batch 0 audible after 2T hops, batch 1's after 3T) and a few more
    n = hops * 1024
    L, R = oracle.synth_audio(n, 4711, True)
    cs = [np.ascontiguousarray(coeffs(k)) for k in range(4)]
    ref = _run(C.CDLL(oracle.ref_path("stream")), 1 << 20, cs, F, T, L, R, chunks)
    got = _run(spleeterrt_amd.load_library(), 4096, cs, F, T, L, R, chunks)
    assert np.all(ref[:, :2 * T * 1024] == 0) and np.all(got[:, :2 * T * 1024] == 0)       # silence for the first 2T hops
    tail_r, tail_g = ref[:, 2 * T * 1024:], got[:, 2 * T * 1024:]
    assert np.abs(tail_r).max() > 1e-3                          # there is signal to compare
    for j in range(8):
        err = np.sqrt(np.mean((tail_g[j] - tail_r[j]) ** 2)) / (np.sqrt(np.mean(tail_r[j] ** 2)) + 1e-30)
        assert err <= 1e-4, "component %d rel rms %g" % (j, err)
    assert np.abs(tail_g - tail_r).max() <= 1e-4 * np.abs(tail_r).max()
    assert np.isfinite(got).all() and np.isfinite(ref).all()
