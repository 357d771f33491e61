import sys, time, torch
sys.path.insert(0, '.')
import spleeterrt_amd as srt
from bench import synth_weights, F, T, STEMS, TILES, HOP
dev = torch.device('cuda', 0)
eng = srt.Engine(F=F, T=T, stem_modes=(1,)*STEMS, variant=srt.VARIANT_VST, max_tiles=TILES, device=dev)
for s in range(STEMS): eng.set_coeff(s, synth_weights(s, dev))
n = TILES * T * HOP
hL = (torch.rand(n) - 0.5).mul_(0.2).pin_memory(); hR = (torch.rand(n) - 0.5).mul_(0.2).pin_memory()
rows = eng.L.srtStftRows(n)
hout = torch.empty((STEMS, 2, eng.L.srtIstftLength(rows))).pin_memory()
dL = torch.empty(n, device=dev); dR = torch.empty(n, device=dev); dout = torch.empty_like(hout, device=dev)
def step():
    dL.copy_(hL, non_blocking=True); dR.copy_(hR, non_blocking=True)
    eng.separate(dL, dR, dout)
    hout.copy_(dout, non_blocking=True)
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 5
for _ in range(K): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print("PCIe-inclusive (pinned host PCM in, pinned host stems out, single stream, no overlap): %.2f ms/step = %.0f frames/s = %.0f x real-time; H2D %.0f MB + D2H %.0f MB per step" % (dt*1e3, rows/dt, rows*1024/44100/dt, 2*n*4/1e6, hout.numel()*4/1e6))

# the same job through the library's own overlapped pipeline: a 4x longer host stream cut into TILES-tile chunks
import numpy as np
K = 8
n2 = K * TILES * T * HOP
L2 = ((np.random.rand(n2) - 0.5) * 0.2).astype(np.float32); R2 = ((np.random.rand(n2) - 0.5) * 0.2).astype(np.float32)
rows2 = eng.L.srtStftRows(n2)
out2 = np.empty((STEMS, 2, eng.L.srtIstftLength(rows2)), np.float32)
eng.separate_host_stream(L2, R2, out=out2)
t0 = time.perf_counter(); eng.separate_host_stream(L2, R2, out=out2); dt = time.perf_counter() - t0
print("PCIe-inclusive, srtSeparateHostStream (pageable numpy in/out registered on entry, 3 streams, double buffers): %.2f ms per %d-tile chunk = %.0f frames/s = %.0f x real-time"
      % (dt * 1e3 / K, TILES, rows2 / dt, rows2 * 1024 / 44100 / dt))
