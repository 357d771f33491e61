"""ctypes binding of include/spleeterrt_amd.h.  Fails loudly if the HIP library is missing — there is no CPU path."""
import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(PKG, "libspleeterrt_amd.so")
MAX_STEMS = 8
VARIANT_EXE, VARIANT_VST = 0, 1
IMPL_MFMA, IMPL_NAIVE = 0, 1
PREC_F32, PREC_F16, PREC_F16X2 = 0, 1, 2
COEFF_FLOATS = 9822725
SPEC_LD = 2052


class EngineError(RuntimeError):
    pass


class _Config(C.Structure):
    _fields_ = [("F", C.c_int), ("T", C.c_int), ("n_stems", C.c_int), ("stem_mode", C.c_int * MAX_STEMS),
                ("oob_weight", C.c_float * MAX_STEMS), ("variant", C.c_int), ("max_tiles", C.c_int), ("impl", C.c_int), ("precision", C.c_int),
                ("ratio_mask", C.c_int), ("batch_invariant", C.c_int)]


class Span(C.Structure):
    """srt_span (include/spleeterrt_amd.h): one rank's share of a stream, as srtRankSpan fills it."""
    _fields_ = [(k, C.c_size_t) for k in ("tile0", "tile1", "sample0", "nsamples", "frames", "rows", "out_offset")]


_lib = None


def load_library():
    """dlopen libspleeterrt_amd.so.  torch is imported first so both share one HIP runtime (same SONAME)."""
    global _lib
    if _lib is not None:
        return _lib
    so = os.environ.get("SPLEETERRT_LIB") or SO           # SPLEETERRT_LIB: the -DSRT_TUNING measurement build (scripts/gpu_tune.sh)
    if not os.path.exists(so):
        raise EngineError("%s not built: run `python -m spleeterrt_amd.build` (needs hipcc); no CPU fallback exists" % so)
    import torch  # noqa: F401  (loads libamdhip64 before our library resolves it)
    L = C.CDLL(so)
    vp, f32p = C.c_void_p, C.c_void_p
    L.srtCreate.argtypes = [C.POINTER(_Config), vp, C.POINTER(vp)]
    L.srtDestroy.argtypes = [vp]
    L.srtDestroy.restype = None
    L.srtLastError.restype = C.c_char_p
    L.srtCoeffBytes.restype = C.c_size_t
    L.srtSetCoeffHost.argtypes = [vp, C.c_int, vp]
    L.srtSetCoeffDevice.argtypes = [vp, C.c_int, vp]
    L.srtSetCoeffFp16Host.argtypes = [vp, C.c_int, vp]
    L.srtGetCoeffHost.argtypes = [vp, C.c_int, vp]
    L.srtForward.argtypes = [vp, f32p, C.c_int, f32p]
    L.srtForwardStems.argtypes = [vp, f32p, C.c_int, f32p, C.c_int, C.c_int]
    L.srtRatioMask.argtypes = [vp, f32p, C.c_int]
    L.srtSeparateCli.argtypes = [vp, f32p, f32p, C.c_size_t, C.c_int, f32p]
    L.srtSeparateCliHost.argtypes = [vp, vp, vp, C.c_size_t, C.c_int, vp]
    L.srtSeparateHostStream.argtypes = [vp, vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, vp]
    L.srtSeparateHostStreamEx.argtypes = [vp, vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, vp, C.c_uint]
    for fn in (L.srtStftRows, L.srtStftFrames, L.srtIstftLength):
        fn.restype = C.c_size_t
        fn.argtypes = [C.c_size_t]
    L.srtStft.argtypes = [vp, f32p, f32p, C.c_size_t, f32p, f32p]
    L.srtIstft.argtypes = [vp, f32p, C.c_size_t, f32p, f32p]
    L.srtSeparate.argtypes = [vp, f32p, f32p, C.c_size_t, f32p]
    L.srtStftEx.argtypes = [vp, f32p, f32p, C.c_size_t, C.c_size_t, C.c_size_t, f32p, f32p]
    L.srtSeparateEx.argtypes = [vp, f32p, f32p, C.c_size_t, C.c_size_t, C.c_size_t, f32p]
    L.srtCopyTensor.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, vp, C.c_size_t]
    L.srtSetGraphMode.argtypes = [vp, C.c_int]
    L.srtPrepareForward.argtypes = [vp, f32p, C.c_int, f32p]
    L.srtReleaseStaging.argtypes = [vp]
    L.srtSetTiming.argtypes = [vp, C.c_int]
    L.srtGetTiming.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_float), C.c_int]
    L.srtGetTimingKernels.argtypes = [vp, C.c_char_p, C.c_size_t]
    # multi-device host driver (csrc/srt_multi.hip)
    L.srtRankSpan.argtypes = [C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(Span)]
    L.srtMultiCreate.argtypes = [C.POINTER(_Config), C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.srtMultiDestroy.argtypes = [vp]
    L.srtMultiDestroy.restype = None
    L.srtMultiSetCoeffHost.argtypes = [vp, C.c_int, vp]
    L.srtMultiSetCoeffFp16Host.argtypes = [vp, C.c_int, vp]
    L.srtMultiSeparateHost.argtypes = [vp, vp, vp, C.c_size_t, vp, C.c_uint]
    L.srtMultiSeparateCliHost.argtypes = [vp, vp, vp, C.c_size_t, C.c_int, vp]
    L.srtMultiInfo.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.srtMultiEngine.argtypes = [vp, C.c_int]
    L.srtMultiEngine.restype = vp
    L.srtMultiBenchResident.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    _lib = L
    return L


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class Engine:
    """One engine per (device, stream): nstems sub-networks evaluated over batches of T x F spectrogram tiles."""

    def __init__(self, F=1024, T=256, stem_modes=(1, 1, 1, 1), oob_weights=None, variant=VARIANT_EXE, max_tiles=1,
                 impl=IMPL_MFMA, device=None, precision=PREC_F32, ratio_mask=False, batch_invariant=False):
        import torch
        if not torch.cuda.is_available():
            raise EngineError("no GPU visible: spleeterrt_amd has no CPU path")
        self.torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        torch.cuda.set_device(self.device)
        self.L = load_library()
        self.F, self.T, self.S, self.max_tiles, self.variant = F, T, len(stem_modes), max_tiles, variant
        cfg = _Config()
        cfg.F, cfg.T, cfg.n_stems, cfg.variant, cfg.max_tiles, cfg.impl = F, T, self.S, variant, max_tiles, impl
        cfg.precision = precision
        cfg.ratio_mask = int(bool(ratio_mask))
        cfg.batch_invariant = int(bool(batch_invariant))
        for i, m in enumerate(stem_modes):
            cfg.stem_mode[i] = int(m)
            cfg.oob_weight[i] = 0.1 if oob_weights is None else float(oob_weights[i])
        self.stream = torch.cuda.current_stream(self.device)
        h = C.c_void_p()
        self._chk(self.L.srtCreate(C.byref(cfg), C.c_void_p(self.stream.cuda_stream), C.byref(h)))
        self.h = h

    def _chk(self, rc):
        if rc < 0:
            raise EngineError("libspleeterrt_amd: %s (rc=%d)" % (self.L.srtLastError().decode(), rc))
        return rc

    def close(self):
        if getattr(self, "h", None):
            self.L.srtDestroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights
    def set_coeff(self, stem, coeff):
        """coeff: numpy float32[9822725] (host) or a CUDA float32 tensor (device), spleeterCoeff layout."""
        import numpy as np
        if isinstance(coeff, np.ndarray):
            a = np.ascontiguousarray(coeff, np.float32)
            assert a.size == COEFF_FLOATS
            self._chk(self.L.srtSetCoeffHost(self.h, stem, C.c_void_p(a.ctypes.data)))
        else:
            assert coeff.is_cuda and coeff.numel() == COEFF_FLOATS and coeff.dtype == self.torch.float32
            self._chk(self.L.srtSetCoeffDevice(self.h, stem, _ptr(coeff.contiguous())))

    def set_coeff_fp16(self, stem, halfs):
        import numpy as np
        a = np.ascontiguousarray(halfs, np.uint16)
        assert a.size == COEFF_FLOATS
        self._chk(self.L.srtSetCoeffFp16Host(self.h, stem, C.c_void_p(a.ctypes.data)))

    def get_coeff(self, stem):
        """the fp32 blob the engine holds for a sub-network (after set_coeff_fp16: the expanded container)"""
        import numpy as np
        a = np.empty(COEFF_FLOATS, np.float32)
        self._chk(self.L.srtGetCoeffHost(self.h, stem, C.c_void_p(a.ctypes.data)))
        return a

    # ---- stages (all tensors live on self.device)
    def forward(self, mag, masks=None):
        """mag [ntiles,2,T,F] -> masks [S,ntiles,2,T,F]"""
        t = self.torch
        nt = mag.shape[0]
        assert mag.is_cuda and mag.dtype == t.float32 and tuple(mag.shape[1:]) == (2, self.T, self.F)
        mag = mag.contiguous()
        if masks is None:
            masks = t.empty((self.S, nt, 2, self.T, self.F), device=self.device, dtype=t.float32)
        self._chk(self.L.srtForward(self.h, _ptr(mag), nt, _ptr(masks)))
        return masks

    def forward_stems(self, mag, masks, stem0, nstems):
        """sub-networks [stem0, stem0+nstems) only; masks keeps the all-stem shape [S,ntiles,2,T,F]"""
        self._chk(self.L.srtForwardStems(self.h, _ptr(mag.contiguous()), mag.shape[0], _ptr(masks), stem0, nstems))
        return masks

    def ratio_mask(self, masks):
        """in place: m_s <- (m_s^2 + eps/S) / (sum_j m_j^2 + eps) across the stem axis"""
        assert masks.is_contiguous() and masks.shape[0] == self.S
        self._chk(self.L.srtRatioMask(self.h, _ptr(masks), masks.shape[1]))
        return masks

    def stft(self, L, R, want_mag=True):
        """planar PCM -> (spec [2,rows,2052,2], mag [ntiles,2,T,F] or None)"""
        t = self.torch
        n = L.numel()
        rows = self.L.srtStftRows(n)
        nt = (rows + self.T - 1) // self.T
        spec = t.empty((2, rows, SPEC_LD, 2), device=self.device, dtype=t.float32)
        mag = t.empty((nt, 2, self.T, self.F), device=self.device, dtype=t.float32) if want_mag else None
        self._chk(self.L.srtStft(self.h, _ptr(L.contiguous()), _ptr(R.contiguous()), n, _ptr(spec), _ptr(mag)))
        return spec, mag

    def istft(self, spec, masks=None):
        """spec [2,rows,2052,2], masks [S,ntiles,2,T,F] or None -> out [S,2,rows*1024+3072]"""
        t = self.torch
        rows = spec.shape[1]
        out = t.empty((self.S, 2, self.L.srtIstftLength(rows)), device=self.device, dtype=t.float32)
        self._chk(self.L.srtIstft(self.h, _ptr(spec), rows, _ptr(masks), _ptr(out)))
        return out

    def separate(self, L, R, out=None):
        """whole path: planar PCM [n] x2 -> stems [S,2,rows*1024+3072]"""
        t = self.torch
        n = L.numel()
        rows = self.L.srtStftRows(n)
        if out is None:
            out = t.empty((self.S, 2, self.L.srtIstftLength(rows)), device=self.device, dtype=t.float32)
        self._chk(self.L.srtSeparate(self.h, _ptr(L), _ptr(R), n, _ptr(out)))
        return out

    def separate_cli(self, L, R, stems):
        """the offline CLI's flow (main.c:776-798 / 845-928): -> [stems,2,len] = (Vocal, Accompaniment) or (Drum, Vocal, Accompaniment)"""
        t = self.torch
        n = L.numel()
        out = t.empty((stems, 2, self.L.srtIstftLength(self.L.srtStftRows(n))), device=self.device, dtype=t.float32)
        self._chk(self.L.srtSeparateCli(self.h, _ptr(L.contiguous()), _ptr(R.contiguous()), n, stems, _ptr(out)))
        return out

    def separate_cli_host(self, L, R, stems, keep_staging=False):
        """the CLI flow from host buffers (numpy float32), any length: one resident batch when the file fits max_tiles tiles, otherwise
        chunk by chunk (srtSeparateCliHost) -> numpy [stems,2,len].  The call's device staging (whole-file PCM + outputs for a file that fits, O(file)
        bytes of HBM) is released afterwards unless keep_staging=True (a caller that separates file after file keeps it to avoid re-allocating)."""
        import numpy as np
        L = np.ascontiguousarray(L, np.float32)
        R = np.ascontiguousarray(R, np.float32)
        assert L.size == R.size
        out = np.empty((stems, 2, self.L.srtIstftLength(self.L.srtStftRows(L.size))), np.float32)
        try:
            self._chk(self.L.srtSeparateCliHost(self.h, C.c_void_p(L.ctypes.data), C.c_void_p(R.ctypes.data), L.size, stems, C.c_void_p(out.ctypes.data)))
        except EngineError:
            if not keep_staging:
                self.L.srtReleaseStaging(self.h)              # best effort: the separation's own error is the one to report
            raise
        if not keep_staging:
            self._chk(self.L.srtReleaseStaging(self.h))
        return out

    def separate_host_stream(self, L, R, frames=None, rows=None, out=None, pinned=False):
        """host PCM of any length -> host stems [S,2,rows*1024+3072]; chunks of max_tiles tiles with the PCIe copies
        overlapped with compute (srtSeparateHostStreamEx).  L, R, out: contiguous float32 numpy arrays or CPU torch
        tensors; pinned=True promises they are page-locked already (torch pin_memory), so nothing is registered per call."""
        import numpy as np

        def host(a):
            if hasattr(a, "data_ptr"):                           # CPU torch tensor (e.g. pin_memory=True)
                assert not a.is_cuda and a.dtype == self.torch.float32 and a.is_contiguous()
                return a, a.data_ptr(), a.numel()
            a = np.ascontiguousarray(a, np.float32)
            return a, a.ctypes.data, a.size
        L, pL, n = host(L)
        R, pR, nR = host(R)
        assert n == nR
        rows = self.L.srtStftRows(n) if rows is None else rows
        frames = self.L.srtStftFrames(n) if frames is None else frames
        shape = (self.S, 2, self.L.srtIstftLength(rows))
        ret = None
        if out is None:
            if pinned:                                           # keep the promise for the output too
                out = self.torch.empty(shape, dtype=self.torch.float32, pin_memory=True)
                ret = out.numpy()
            else:
                out = np.empty(shape, np.float32)
        out, pO, no = host(out)
        assert no == shape[0] * shape[1] * shape[2]
        self._chk(self.L.srtSeparateHostStreamEx(self.h, C.c_void_p(pL), C.c_void_p(pR), n, frames, rows, C.c_void_p(pO),
                                                 1 if pinned else 0))
        return out if ret is None else ret

    def separate_ex(self, L, R, frames, rows, out=None):
        """explicit-geometry form used by spleeterrt_amd.stream for tile ranges of a longer stream"""
        t = self.torch
        if out is None:
            out = t.empty((self.S, 2, self.L.srtIstftLength(rows)), device=self.device, dtype=t.float32)
        self._chk(self.L.srtSeparateEx(self.h, _ptr(L.contiguous()), _ptr(R.contiguous()), L.numel(), frames, rows, _ptr(out)))
        return out

    # ---- debug / measurement
    def tensor(self, name, stem, tile):
        import numpy as np
        lvl = int(name[-1])
        if name.startswith("up"):
            co = (256, 128, 64, 32, 16, 1)[lvl - 1]
            sh = (co, self.T >> (6 - lvl), self.F >> (6 - lvl))
        else:
            co = (16, 32, 64, 128, 256, 512)[lvl - 1]
            sh = (co, self.T >> lvl, self.F >> lvl)
        a = np.empty(sh, np.float32)
        self._chk(self.L.srtCopyTensor(self.h, name.encode(), stem, tile, C.c_void_p(a.ctypes.data), a.size))
        return a

    def set_graph_mode(self, on=True):
        """replay srtForward / srtSeparate as captured hipGraphs when called again with the same tensors (needs a non-default stream)"""
        self._chk(self.L.srtSetGraphMode(self.h, int(on)))

    def prepare_forward(self, mag, masks):
        """allocate / capture everything forward(mag, masks) would do lazily (runs the networks once into masks)"""
        self._chk(self.L.srtPrepareForward(self.h, _ptr(mag), mag.shape[0], _ptr(masks)))

    def release_staging(self):
        self._chk(self.L.srtReleaseStaging(self.h))

    def set_timing(self, on=True):
        self._chk(self.L.srtSetTiming(self.h, int(on)))

    def get_timing(self, max_entries=65536):
        names = C.create_string_buffer(max_entries * 8)
        ms = (C.c_float * max_entries)()
        n = self._chk(self.L.srtGetTiming(self.h, names, len(names), ms, max_entries))
        nm = names.value.decode().split(",")[:n]
        return list(zip(nm, list(ms[:n])))

    def get_timing_kernels(self, max_entries=65536):
        """[(launch name, kernel symbol that ran it)] for the launches recorded since set_timing(True)"""
        tim = self.get_timing(max_entries)
        buf = C.create_string_buffer(len(tim) * 160 + 16)
        n = self._chk(self.L.srtGetTimingKernels(self.h, buf, len(buf)))
        ks = buf.value.decode().split(";")[:n]
        return [(name, k) for (name, _), k in zip(tim, ks)]
