#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (oracle/_ref, built from /root/reference by oracle/Makefile).

Run in the build container only:  python tests/golden/gen_golden.py
The fixtures hold inputs' seeds, sub-sampled outputs and full-tensor statistics — data, never reference source.
Weights are not stored (39 MB each): both sides regenerate them from the seed (oracle.synth_coeff).
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle as O  # noqa: E402


def stats(a):
    a64 = a.astype(np.float64).ravel()
    w = np.cos(np.arange(a64.size) * 0.61803398875)            # position-sensitive checksum
    return np.array([a64.sum(), (a64 ** 2).sum(), (a64 * w).sum(), a64.min(), a64.max()])


def sub(a, step):
    return np.ascontiguousarray(a.ravel()[::step])


def mag_input(ntiles, T, F, seed):
    x = np.abs(O.lcg(seed, ntiles * 2 * T * F, 6.0)).reshape(ntiles, 2, T, F)
    x[:, :, ::7, ::13] *= 8.0
    return np.ascontiguousarray(x, np.float32)


def main():
    assert O.ref_path("exe"), "oracle/_ref missing: run `make -C oracle ref` where /root/reference exists"
    out = {}
    # ---- network forward, both flavours, both activation modes, T=64 F=512
    T, F = 64, 512
    x = mag_input(1, T, F, 4242)[0]
    for flavour, variant in (("exe", "exe"), ("vst", "vst")):
        for stem, mode in ((0, 0), (1, 1)):
            net = O.RefNet(O.synth_coeff(stem), F, T, mode, flavour)
            y = net(x)
            net.close()
            key = "fwd_%s_stem%d_mode%d" % (variant, stem, mode)
            out[key + "_sub"] = sub(y, 7)
            out[key + "_stats"] = stats(y)
    # ---- per-primitive activations from the reference's exported symbols
    L = C.CDLL(O.ref_path("exe"))
    xs = np.linspace(-8.5, 8.5, 4001).astype(np.float32)
    for name in ("fastSigmoid", "ELU", "leakyReLU", "ReLU"):
        fn = getattr(L, name)
        fn.restype = C.c_float
        fn.argtypes = [C.c_float]
        out["act_" + name] = np.array([fn(float(v)) for v in xs], np.float32)
    out["act_x"] = xs
    # ---- STFT / iSTFT on a ragged-length toned signal
    n = 4096 * 3 + 8192 + 777
    Lc, Rc = O.synth_audio(n, 777, True)
    st = O.RefSTFT(1)
    re, im = st.stft(Lc, Rc)
    out["stft_n"] = np.array([n])
    out["stft_re_sub"] = sub(re[:, :, :2049], 5); out["stft_im_sub"] = sub(im[:, :, :2049], 5)
    out["stft_re_stats"] = stats(re); out["stft_im_stats"] = stats(im)
    y = st.istft(re, im)
    out["istft_sub"] = sub(y, 3); out["istft_stats"] = stats(y)
    st.close()
    # ---- Hartley transform of one seeded vector (input given in natural order; the codelet wants it bit-reversed)
    L.DFT4096.argtypes = [np.ctypeslib.ndpointer(np.float32), np.ctypeslib.ndpointer(np.float32)]
    a = O.lcg(99, 4096, 2.0)
    t = O.tables()
    rev = np.array(t.rev[:], np.int64)
    br = np.empty(4096, np.float32); br[rev] = a
    sine = np.array(t.sine[:], np.float32)
    L.DFT4096(br, sine)
    out["fht_out"] = br
    np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"), **out)
    print("wrote", os.path.join(HERE, "reference_vectors.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
