#!/usr/bin/env python3
"""BASELINE configs[1]: 2-stem, one 256-frame tile end-to-end (STFT + U-Nets + mask + iSTFT) latency on one MI355X."""
import sys, time, torch
sys.path.insert(0, '.')
import spleeterrt_amd as srt
from bench import synth_weights
prec = {"f32": srt.PREC_F32, "f16": srt.PREC_F16, "f16x2": srt.PREC_F16X2}[sys.argv[1] if len(sys.argv) > 1 else "f32"]
dev = torch.device('cuda', 0)
eng = srt.Engine(F=1024, T=256, stem_modes=(0, 1), variant=srt.VARIANT_EXE, max_tiles=1, device=dev, precision=prec)
for s in range(2): eng.set_coeff(s, synth_weights(s, dev))
n = 256 * 1024
L = (torch.rand(n, device=dev) - 0.5) * 0.2; R = (torch.rand(n, device=dev) - 0.5) * 0.2
out = eng.separate(L, R)
for _ in range(3): eng.separate(L, R, out)
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 20
for _ in range(K): eng.separate(L, R, out)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
eng.set_timing(True); eng.separate(L, R, out); tim = eng.get_timing()
print("C2 latency (2 stems, 1 tile of 256x1024, %s): %.3f ms per tile = %.0f x real-time" % (sys.argv[1] if len(sys.argv) > 1 else "f32", dt * 1e3, 256 * 1024 / 44100 / dt))
agg = {}
[agg.__setitem__(k, agg.get(k, 0.0) + v) for k, v in tim]
print({k: round(v, 3) for k, v in agg.items()})
