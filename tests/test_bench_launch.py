"""`bench.py --gpus N` is binding: it starts the N ranks itself when no launcher did, refuses to print a line for fewer
devices than asked, and refuses a launcher world that differs from N.  CPU-only here: the rendezvous half (`--rendezvous-only`, hidden flag)
joins a real gloo world of 2 through the same torch.distributed.run command the GPU path uses; the GPU halves are in tests/test_rccl.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = dict(os.environ, **extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    return env


def _cpu_only_env():
    return _env(HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")     # also on a GPU box: no device visible


def test_more_gpus_than_devices_fails_loudly():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=_cpu_only_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2 but 0 GPU(s) visible" in r.stderr, r.stderr[-800:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]                    # and no JSON line at all
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--host", "native", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=_cpu_only_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 2 --host native but 0 GPU(s) visible" in r.stderr, r.stderr[-800:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_launcher_world_must_equal_gpus():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8"], cwd=ROOT, env=dict(_env(), RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="1"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 8 but the launcher started WORLD_SIZE=2" in r.stderr, r.stderr[-800:]


def test_gpus_2_without_a_launcher_starts_two_ranks():
    """python bench.py --gpus 2 (no RANK/WORLD_SIZE in the environment) re-launches itself under torch.distributed.run with two ranks, which meet
    in a process group (gloo here) and all-reduce: rank_sum 1 + 2 = 3 proves both were there."""
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--rendezvous-only"], cwd=ROOT, env=_cpu_only_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d == {"rendezvous": True, "world": 2, "n_gpus": 2, "backend": "gloo", "rank_sum": 3.0}


def test_up6_streamed_f16_kernel_is_priced_against_the_fp16_peak():
    """the fp16 modes run up6 as srt_up6_stream_kernel<.., .., .., true> (v_mfma_f32_32x32x16_f16)."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.is_f16_kernel("srt_up6_stream_kernel<64, 2, 0, true>") and bench.mfma_peak("srt_up6_stream_kernel<64, 2, 0, true>") == 2500.0
    assert not bench.is_f16_kernel("srt_up6_stream_kernel<64, 2, 0, false>") and bench.mfma_peak("srt_up6_stream_kernel<64, 2, 0, false>") == 157.3
    assert bench.layer_bytes("down1", "f32", False, 5) < bench.layer_bytes("down1", "f32", False, 4)      # the magnitudes are shared by the stems
