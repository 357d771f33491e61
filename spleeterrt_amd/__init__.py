"""spleeterrt_amd — MI355X-native engine for SpleeterRT's STFT -> U-Net mask -> iSTFT hot path.

The product is the C-ABI shared library `libspleeterrt_amd.so` (hand-written HIP for gfx950, see csrc/ and
include/*.h).  This package only holds the build script and a thin ctypes binding used by tests and bench.py;
PyTorch appears solely as the owner of device memory / streams / torch.distributed.
"""
from .capi import Engine, EngineError, load_library, VARIANT_EXE, VARIANT_VST, IMPL_MFMA, IMPL_NAIVE, PREC_F32, PREC_F16, PREC_F16X2  # noqa: F401
