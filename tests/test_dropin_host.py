"""Drop-in proof (GPU): the SAME plain-C host program (host/offline_main.c, written only against the reference's
spleeter.h + stftFix.h API) is linked once against libspleeterrt_amd.so and once against the real reference
(oracle/_ref) and must produce the same stems.  Also exercises the Python view of the same C entry points."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "host")


def _rel_rms(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2)) / (np.sqrt(np.mean(b ** 2)) + 1e-30))


@pytest.fixture(scope="module")
def workdir(tmp_path_factory, oracle):
    d = tmp_path_factory.mktemp("dropin")
    # fp16 container with 2 sub-nets: net[0] drum (ELU), net[1] vocal (LeakyReLU/ReLU)  (main.c:759-760)
    h = np.concatenate([oracle.synth_coeff_fp16(1), oracle.synth_coeff_fp16(0)])
    h.tofile(d / "weights.f16")
    n = 44100 * 3 + 123
    L, R = oracle.synth_audio(n, 777, True)
    np.stack([L, R], 1).astype(np.float32).tofile(d / "in.f32")
    return d


@pytest.mark.parametrize("stems", [2, 3])
def test_same_host_program_two_backends(workdir, oracle, stems):
    amd, ref = os.path.join(HOST, "offline_amd"), os.path.join(HOST, "offline_ref")
    if not os.path.exists(amd):
        subprocess.check_call(["make", "-s", "-C", HOST, "offline_amd"])
    if not os.path.exists(ref) or oracle.ref_path("exe") is None:
        pytest.skip("reference build (oracle/_ref, host/offline_ref) not present")
    env = dict(os.environ, SPLEETERRT_VARIANT="exe")
    for exe, tag in ((amd, "amd"), (ref, "ref")):
        subprocess.check_call([exe, "64", "512", str(stems), str(workdir / "weights.f16"), str(workdir / "in.f32"),
                               str(workdir / ("out%d_%s" % (stems, tag)))], env=env)
    names = ["Vocal", "Accompaniment"] + (["Drum"] if stems == 3 else [])
    for nm in names:
        a = np.fromfile(workdir / ("out%d_amd_%s.f32" % (stems, nm)), np.float32)
        r = np.fromfile(workdir / ("out%d_ref_%s.f32" % (stems, nm)), np.float32)
        assert a.shape == r.shape and a.size == 2 * (44100 * 3 + 123)
        assert _rel_rms(a, r) <= 1e-4, "%s: rel rms %g" % (nm, _rel_rms(a, r))
        assert np.abs(a - r).max() <= 1e-4 * np.abs(r).max()


def test_tile_api_via_ctypes(oracle, coeffs):
    """initSpleeter / processSpleeter / getMaskPtr with host pointers, y aliasing the mask buffer (main.c:453,472)."""
    import spleeterrt_amd
    L = spleeterrt_amd.load_library()
    os.environ["SPLEETERRT_VARIANT"] = "vst"
    T, F = 64, 512
    L.allocateSpleeterStr.restype = C.c_void_p
    nn = C.c_void_p(L.allocateSpleeterStr())
    c = np.ascontiguousarray(coeffs(1))
    L.initSpleeter.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
    L.initSpleeter(nn, F, T, 1, c.ctypes.data)
    mask = C.POINTER(C.c_float)()
    L.getMaskPtr.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_float))]
    L.getMaskPtr(nn, C.byref(mask))
    x = np.abs(oracle.lcg(8, 2 * T * F, 6.0)).reshape(2, T, F).astype(np.float32)
    L.processSpleeter.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.processSpleeter(nn, x.ctypes.data, mask)
    y = np.ctypeslib.as_array(mask, shape=(2, T, F)).copy()
    assert np.abs(y - oracle.forward(c, x, 1, oracle.VARIANT_VST)).max() <= 2e-4
    L.freeSpleeter.argtypes = [C.c_void_p]
    L.freeSpleeter(nn)
    C.CDLL(None).free.argtypes = [C.c_void_p]
    C.CDLL(None).free(nn)
    os.environ.pop("SPLEETERRT_VARIANT")


def _write_wav16(path, L, R, rate=44100):
    import struct
    pcm = np.clip(np.round(np.stack([L, R], 1) * 32768.0), -32768, 32767).astype("<i2")
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + pcm.nbytes) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 2, rate, rate * 4, 4, 16))
        f.write(b"data" + struct.pack("<I", pcm.nbytes) + pcm.tobytes())
    return pcm.astype(np.float32) / 32768.0


def _read_wav_f32(path):
    b = open(path, "rb").read()
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE"
    i = b.index(b"data")
    n = int.from_bytes(b[i + 4:i + 8], "little")
    return np.frombuffer(b[i + 8:i + 8 + n], "<f4").reshape(-1, 2)


@pytest.mark.parametrize("stems", [2, 3])
def test_cli_program_matches_reference_linked_harness(workdir, oracle, stems):
    """host/spleeterrt_cli (reference command line, WAV in / float32 WAV out, device-resident flow) against the harness
    linked to the real reference on the same 16-bit PCM clip."""
    cli, ref = os.path.join(HOST, "spleeterrt_cli"), os.path.join(HOST, "offline_ref")
    if not os.path.exists(cli):
        subprocess.check_call(["make", "-s", "-C", HOST, "spleeterrt_cli"])
    if not os.path.exists(ref) or oracle.ref_path("exe") is None:
        pytest.skip("reference build (oracle/_ref, host/offline_ref) not present")
    n = 44100 * 2 + 77
    L, R = oracle.synth_audio(n, 555, True)
    q = _write_wav16(workdir / "clip.wav", L * 4.0, R * 4.0)
    q.astype(np.float32).tofile(workdir / "clip.f32")
    env = dict(os.environ, SPLEETERRT_VARIANT="exe")
    out = subprocess.check_output([cli, "2", "64", "512", str(stems), str(workdir / "clip.wav"), str(workdir / "weights.f16")],
                                  cwd=workdir, env=env).decode()
    assert "Saving file -> clip.wav_Vocal.wav" in out
    subprocess.check_call([ref, "64", "512", str(stems), str(workdir / "weights.f16"), str(workdir / "clip.f32"),
                           str(workdir / ("cliref%d" % stems))], env=env)
    for nm in ["Vocal", "Accompaniment"] + (["Drum"] if stems == 3 else []):
        a = _read_wav_f32(workdir / ("clip.wav_%s.wav" % nm))
        r = np.fromfile(workdir / ("cliref%d_%s.f32" % (stems, nm)), np.float32).reshape(-1, 2)
        assert a.shape == r.shape == (n, 2)
        assert _rel_rms(a, r) <= 1e-4, "%s: rel rms %g" % (nm, _rel_rms(a, r))
        assert np.abs(a - r).max() <= 1e-4 * np.abs(r).max()


@pytest.mark.parametrize("stems", [2, 3])
def test_baseline_config0_ten_second_clip(workdir, oracle, stems):
    """BASELINE configs[0] at its named size (and its 3-output sibling): 2-stem separation of a 10 s 44.1 kHz stereo WAV at timeStep 256 / bin limit 1024
    (the reference CLI's `2 256 1024 2 file`), spleeterrt_cli on the GPU against the harness linked to the real reference."""
    cli, ref = os.path.join(HOST, "spleeterrt_cli"), os.path.join(HOST, "offline_ref")
    if not os.path.exists(cli):
        subprocess.check_call(["make", "-s", "-C", HOST, "spleeterrt_cli"])
    if not os.path.exists(ref) or oracle.ref_path("exe") is None:
        pytest.skip("reference build (oracle/_ref, host/offline_ref) not present")
    n = 441000
    L, R = oracle.synth_audio(n, 1234, True)
    q = _write_wav16(workdir / "ten.wav", L * 4.0, R * 4.0)
    q.astype(np.float32).tofile(workdir / "ten.f32")
    env = dict(os.environ, SPLEETERRT_VARIANT="exe")
    subprocess.check_call([cli, "1", "256", "1024", str(stems), str(workdir / "ten.wav"), str(workdir / "weights.f16")], cwd=workdir, env=env,
                          stdout=subprocess.DEVNULL)
    subprocess.check_call([ref, "256", "1024", str(stems), str(workdir / "weights.f16"), str(workdir / "ten.f32"), str(workdir / "tenref")], env=env)
    for nm in ["Vocal", "Accompaniment"] + (["Drum"] if stems == 3 else []):
        a = _read_wav_f32(workdir / ("ten.wav_%s.wav" % nm))
        r = np.fromfile(workdir / ("tenref_%s.f32" % nm), np.float32).reshape(-1, 2)
        assert a.shape == r.shape == (n, 2)
        assert _rel_rms(a, r) <= 1e-4, "%s: rel rms %g" % (nm, _rel_rms(a, r))
        assert np.abs(a - r).max() <= 1e-4 * np.abs(r).max()


def test_cli_program_walks_long_files_in_chunks(workdir, oracle):
    """The CLI binary with a small engine capacity (SPLEETERRT_MAX_TILES=3: the 10 s clip is 8 tiles of 64 frames -> 3 chunks, the last
    ragged) writes the same three files as with the default capacity (one resident batch): any file length works, as with the
    reference's tile-at-a-time loop (main.c:455-495).  SPLEETERRT_BATCH_INVARIANT=1 pins the kernel choice, so only the overlap-add
    association at the two chunk seams differs."""
    cli = os.path.join(HOST, "spleeterrt_cli")
    subprocess.check_call(["make", "-s", "-C", HOST, "spleeterrt_cli"])
    n = 441000
    L, R = oracle.synth_audio(n, 4321, True)
    _write_wav16(workdir / "long.wav", L * 4.0, R * 4.0)
    outs = {}
    for tag, extra in (("resident", {}), ("chunked", {"SPLEETERRT_MAX_TILES": "3"})):
        d = workdir / tag
        d.mkdir(exist_ok=True)
        env = dict(os.environ, SPLEETERRT_VARIANT="exe", SPLEETERRT_BATCH_INVARIANT="1", **extra)
        txt = subprocess.check_output([cli, "1", "64", "512", "3", str(workdir / "long.wav"), str(workdir / "weights.f16")], cwd=d, env=env).decode()
        assert ("in chunks of 3" in txt) == (tag == "chunked"), txt
        outs[tag] = {nm: _read_wav_f32(d / ("long.wav_%s.wav" % nm)) for nm in ("Drum", "Vocal", "Accompaniment")}
    for nm in outs["resident"]:
        a, r = outs["chunked"][nm], outs["resident"][nm]
        assert a.shape == r.shape == (n, 2)
        assert np.abs(a - r).max() <= 2e-6 * np.abs(r).max(), nm
