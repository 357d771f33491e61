#!/usr/bin/env python3
"""Small-batch latency on one MI355X (the regime of BASELINE configs[1] and of the real-time plugin):
  c2  : 2 stems, ONE 256x1024 tile, PCM -> STFT -> U-Nets -> mask -> iSTFT, everything resident in HBM
  vst : the plugin's batch, 4 stems x one 256x1536 tile, network only (what Spleeter4Stems starts every T hops)
each timed eagerly and as a replayed hipGraph (srtSetGraphMode), plus the per-kernel breakdown and a bit-stability check.
    python scripts/latency_c2.py [f32|f16|f16x2] [--out profiles/r02_latency.json]"""
import json
import sys
import time

import torch

sys.path.insert(0, '.')
import spleeterrt_amd as srt
from bench import synth_weights

prec_name = next((a for a in sys.argv[1:] if a in ("f32", "f16", "f16x2")), "f32")
out_path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
prec = {"f32": srt.PREC_F32, "f16": srt.PREC_F16, "f16x2": srt.PREC_F16X2}[prec_name]
dev = torch.device('cuda', 0)
res = {"precision": prec_name}
side = torch.cuda.Stream(device=dev)                      # graphs need a capturable (non-null) stream


def timed(fn, K=100):
    for _ in range(40):                                   # long enough for the clocks to come back up after the engine set-up
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        fn()
    t_issue = (time.perf_counter() - t0) / K
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K, t_issue


with torch.cuda.stream(side):
    # ---- c2
    eng = srt.Engine(F=1024, T=256, stem_modes=(0, 1), variant=srt.VARIANT_EXE, max_tiles=1, device=dev, precision=prec)
    for s in range(2):
        eng.set_coeff(s, synth_weights(s, dev))
    n = 256 * 1024
    L = (torch.rand(n, device=dev) - 0.5) * 0.2
    R = (torch.rand(n, device=dev) - 0.5) * 0.2
    out = eng.separate(L, R)
    ref = out.clone()
    dt, ti = timed(lambda: eng.separate(L, R, out))
    res["c2_eager_ms"], res["c2_eager_host_issue_ms"] = dt * 1e3, ti * 1e3
    assert torch.equal(out, ref), "c2: not bit-stable run to run"
    eng.set_timing(True); eng.separate(L, R, out); tim = eng.get_timing(); eng.set_timing(False)
    agg = {}
    for k, v in tim:
        agg[k] = agg.get(k, 0.0) + v
    res["c2_kernel_ms"] = {k: round(v, 4) for k, v in agg.items()}
    eng.set_graph_mode(True)
    eng.separate(L, R, out)
    dt, ti = timed(lambda: eng.separate(L, R, out))
    res["c2_graph_ms"], res["c2_graph_host_issue_ms"] = dt * 1e3, ti * 1e3
    assert torch.equal(out, ref), "c2: graph replay differs from the eager result"
    res["c2_x_realtime"] = 256 * 1024 / 44100 / (min(res["c2_eager_ms"], res["c2_graph_ms"]) * 1e-3)
    eng.close()
    # ---- vst batch: 4 stems x 1 tile of 256 x 1536, network only
    eng = srt.Engine(F=1536, T=256, stem_modes=(1, 1, 1, 1), variant=srt.VARIANT_VST, max_tiles=1, device=dev, precision=prec)
    for s in range(4):
        eng.set_coeff(s, synth_weights(s, dev))
    mag = torch.rand((1, 2, 256, 1536), device=dev) * 6.0
    masks = eng.forward(mag)
    ref = masks.clone()
    dt, ti = timed(lambda: eng.forward(mag, masks))
    res["vst_eager_ms"], res["vst_eager_host_issue_ms"] = dt * 1e3, ti * 1e3
    assert torch.equal(masks, ref), "vst: not bit-stable run to run"
    eng.set_timing(True); eng.forward(mag, masks); tim = eng.get_timing(); eng.set_timing(False)
    agg = {}
    for k, v in tim:
        agg[k] = agg.get(k, 0.0) + v
    res["vst_kernel_ms"] = {k: round(v, 4) for k, v in agg.items()}
    eng.set_graph_mode(True)
    eng.forward(mag, masks)
    dt, ti = timed(lambda: eng.forward(mag, masks))
    res["vst_graph_ms"], res["vst_graph_host_issue_ms"] = dt * 1e3, ti * 1e3
    assert torch.equal(masks, ref), "vst: graph replay differs from the eager result"
    eng.close()
print("C2 latency (2 stems, 1 tile of 256x1024, %s): eager %.3f ms, graph %.3f ms per tile = %.0f x real-time; VST batch (4 x 256x1536 forward): eager %.3f ms, graph %.3f ms"
      % (prec_name, res["c2_eager_ms"], res["c2_graph_ms"], res["c2_x_realtime"], res["vst_eager_ms"], res["vst_graph_ms"]))
print(json.dumps(res))
if out_path:
    open(out_path, "w").write(json.dumps(res, indent=1) + "\n")
