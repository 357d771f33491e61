"""The real-time contract of the streaming surface, measured where the plugin's host would measure it (VERDICT r2 missing #4 / next #8).

Reference: LLPAMSProcessNPR never blocks longer than a hop's own transforms, except at the hop that completes a batch of T hops, where
the networks started one batch earlier are joined (VST/Source/Spleeter4Stems.c:351-371); the plugin calls it from the audio callback
with <= 1024 samples (PluginProcessor.cpp:173-181), i.e. once per 23.2 ms at 44.1 kHz.

host/rt_latency.c (plain C over include/Spleeter4Stems.h) drives TWO instances from two host threads at the shipped geometry
F = 1536, T = 256 (PluginProcessor.cpp:124) for 3*T hops each, one hop per call, and times every call.  Bounds (wall time per call,
p99 over the run, each instance): ordinary hops < 2 ms, the T-hop join hops < 5 ms - 9 % and 22 % of the hop period - both with calls
back to back (the GPU never idles; the instances' hop and network streams contend) and paced at the real hop period (the GPU idles
between calls).  The WORST call of every run must stay under the hop period itself (23.2 ms: the contract proper; the recorded runs show
isolated ~1 ms spikes, host scheduling).  Initialisation (weight upload, packing, workspace allocation, hipGraph capture) is timed
separately and is NOT part of any call.  Two more runs drive EIGHT instances at once (eight plugin instances in one DAW on one GPU; nothing
batches their hops across instances - each owns its hop stream and its network stream).  Paced at the real hop period - the situation the contract
is about - every call of every instance, the worst one included, must stay under 5 ms (a fifth of the hop period: a host with its own DSP in the callback
keeps the rest), ordinary hops p99 under 2 ms.  Back to back (an artificial burst no host produces: eight network batches collide with every instance's
hops) p99 must stay under 5 ms and the single worst call under 10 ms (recorded as `worst_call_us`).  The numbers go to gpurun_out/r06_latency.json
(copied to profiles/)."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "host")
F, T = 1536, 256
ORDINARY_P99_US, JOIN_P99_US = 2000.0, 5000.0
EIGHT_PACED_WORST_US = 5000.0                                  # every call of eight paced instances (measured <= 0.5 ms)
EIGHT_BURST_WORST_US = 10000.0                                 # every call of eight instances called back to back (measured <= 2 ms; under half a hop period)
HOP_US = 1024 / 44100 * 1e6


def test_streaming_call_latency_two_instances(tmp_path, coeffs):
    subprocess.check_call(["make", "-s", "-C", HOST, "rt_latency"])
    w = tmp_path / "w4.f32"
    with open(w, "wb") as f:
        for k in range(4):
            np.ascontiguousarray(coeffs(k), np.float32).tofile(f)
    record = {"geometry": {"F": F, "T": T}, "bounds_us": {"ordinary_p99": ORDINARY_P99_US, "join_p99": JOIN_P99_US},
              "hop_period_us": 1024 / 44100 * 1e6, "runs": {}}
    # back to back: 10 T hops = 10 join hops per instance (the join statistics need more than the 3 samples of 3 T hops)
    for tag, pace, hops, ni in (("back_to_back", 0, 10 * T, 2), ("real_time_paced", 23220, T + T // 4, 2), ("eight_instances_back_to_back", 0, 3 * T, 8),
                                ("eight_instances_real_time_paced", 23220, T + T // 4, 8)):
        out = tmp_path / (tag + ".json")
        subprocess.check_call([os.path.join(HOST, "rt_latency"), str(F), str(T), str(hops), str(w), str(pace), str(out), str(ni)], timeout=900)
        r = json.load(open(out))
        record["runs"][tag] = r
        assert len(r["instances"]) == ni
        for i, inst in enumerate(r["instances"]):
            assert inst["init_error"] == "", "%s instance %d came up muted: %s" % (tag, i, inst["init_error"])
            o, j = inst["ordinary_hops"], inst["join_hops"]
            assert o["p50_us"] > 20.0, "%s instance %d: calls return in %.1f us - no GPU work is being done" % (tag, i, o["p50_us"])
            assert j["n"] == hops // T and o["n"] == hops - hops // T
            if ni == 2:
                assert o["p99_us"] < ORDINARY_P99_US, "%s instance %d ordinary hops: %r" % (tag, i, o)
                assert j["p99_us"] < JOIN_P99_US, "%s instance %d join hops: %r" % (tag, i, j)
                assert max(o["max_us"], j["max_us"]) < HOP_US, "%s instance %d: a call took longer than a hop period: %r %r" % (tag, i, o, j)
            elif pace:                                         # eight instances paced like eight plugins in one host: the contract proper, worst call included
                # round 6: 0.37-0.50 ms worst call on every instance (round 5: 8.2-11.1 ms - each instance's FIRST call, loading the hop kernels' code behind
                # seven other instances doing the same; Init pre-warms the hop path now, and the hop stream runs at the device's highest priority)
                assert max(o["max_us"], j["max_us"]) < EIGHT_PACED_WORST_US, "%s instance %d: worst call above %g us: %r %r" % (tag, i, EIGHT_PACED_WORST_US, o, j)
                assert o["p99_us"] < ORDINARY_P99_US, "%s instance %d ordinary hops: %r" % (tag, i, o)
                assert j["p99_us"] < JOIN_P99_US, "%s instance %d join hops: %r" % (tag, i, j)
            else:                                              # eight un-batched instances, no pacing (an artificial burst no host produces)
                assert o["p99_us"] < JOIN_P99_US and j["p99_us"] < JOIN_P99_US, "%s instance %d: %r %r" % (tag, i, o, j)
                worst = max(o["max_us"], j["max_us"])
                r["worst_call_us"] = max(r.get("worst_call_us", 0.0), worst)
                assert worst < EIGHT_BURST_WORST_US, "%s instance %d: worst call above %g us (round 6 measured <= 2 ms): %r %r" % (tag, i, EIGHT_BURST_WORST_US, o, j)
            if hops > 2 * T:
                assert inst["output_peak"] > 1e-4              # the stream is past its 2T hops of silence: real audio came out
    print("latency:", json.dumps(record["runs"]))
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        json.dump(record, open(os.path.join(d, "r06_latency.json"), "w"), indent=1)
