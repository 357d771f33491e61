"""CPU tests: the C-ABI library loads without a GPU, exports every symbol the public headers declare, answers the
pure-host geometry calls, and refuses (loudly, with an error code) to create an engine without a HIP device."""
import ctypes as C
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")


@pytest.fixture(scope="module")
def lib():
    from spleeterrt_amd import build as b
    so = b.build(verbose=False)
    import spleeterrt_amd
    return spleeterrt_amd.load_library(), so


def _declared():
    names = set()
    for h in os.listdir(INC):
        txt = "\n".join(ln for ln in open(os.path.join(INC, h)).read().splitlines() if not ln.lstrip().startswith("#"))
        names |= set(re.findall(r"(?:SRT_API|SPLEETER_API|STFT_API|S4S_API)\s+[\w\s\*]*?\b(\w+)\s*\(", txt))
    return names


def test_every_declared_symbol_is_exported(lib):
    L, so = lib
    declared = _declared()
    assert {"srtCreate", "srtForward", "srtSeparate", "initSpleeter", "processSpleeter", "getMaskPtr", "InitSTFT", "stft", "istft", "Spleeter4StemsInit", "Spleeter4StemsProcessSamples", "Spleeter4StemsFree"} <= declared
    exported = {ln.split()[-1] for ln in subprocess.check_output(["nm", "-D", "--defined-only", so], text=True).splitlines()}
    assert declared <= exported, declared - exported
    # and nothing private leaks: only the three API families are visible
    assert all(n.startswith("srt") or n in declared for n in exported if not n.startswith("_")), exported


def test_sizes_and_geometry(lib):
    L, _ = lib
    L.getCoeffSize.restype = C.c_size_t
    assert L.srtCoeffBytes() == L.getCoeffSize() == 39290900
    # main.c:762-766 padding: n = 4096*ceil(441000/4096) + 8192 -> 440 rows, 437 transformed frames
    n = 4096 * 108 + 8192
    assert L.srtStftRows(n) == 440 and L.srtStftFrames(n) == 437 and L.srtIstftLength(440) == 440 * 1024 + 3072
    assert L.srtStftFrames(4096) == 1 and L.srtStftRows(4097) == 5


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import spleeterrt_amd as srt
    with pytest.raises(srt.EngineError):
        srt.Engine()
    from spleeterrt_amd.capi import _Config
    cfg = _Config()
    cfg.F, cfg.T, cfg.n_stems, cfg.max_tiles = 512, 64, 1, 1
    h = C.c_void_p()
    L, _ = lib
    assert L.srtCreate(C.byref(cfg), None, C.byref(h)) < 0 and b"no HIP device" in L.srtLastError()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "spleeterrt_amd")
    for dp, _, files in os.walk(os.path.join(pkg, "csrc")):
        for f in files:
            assert "oracle" not in open(os.path.join(dp, f)).read().lower(), f
    for f in ("capi.py", "build.py", "stream.py", "__init__.py"):
        txt = open(os.path.join(pkg, f)).read()
        assert "pyoracle" not in txt and "liboracle" not in txt


def test_cli_argument_and_input_validation(tmp_path):
    """host/spleeterrt_cli rejects what it cannot do BEFORE touching the GPU: usage, non-WAVE input, wrong sample rate,
    missing weights (it has no resampler / FLAC / MP3 decoder and no embedded model — see its header)."""
    import struct
    import subprocess
    host = os.path.join(ROOT, "host")
    cli = os.path.join(host, "spleeterrt_cli")
    if not os.path.exists(cli):
        subprocess.check_call(["make", "-s", "-C", host, "spleeterrt_cli"])
    env = {k: v for k, v in os.environ.items() if k != "SPLEETERRT_WEIGHTS"}
    r = subprocess.run([cli], capture_output=True, env=env)
    assert r.returncode != 0 and b"spawnNthreads timeStep analyseBinLimit stems" in r.stdout
    bad = tmp_path / "x.flac"
    bad.write_bytes(b"fLaC" + bytes(64))
    w = tmp_path / "w.f16"
    w.write_bytes(b"")
    r = subprocess.run([cli, "1", "64", "512", "2", str(bad), str(w)], capture_output=True, env=env)
    assert r.returncode != 0 and b"not a RIFF/WAVE" in r.stderr
    wav = tmp_path / "a.wav"
    pcm = bytes(4 * 100)
    wav.write_bytes(b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 2, 48000, 48000 * 4, 4, 16)
                    + b"data" + struct.pack("<I", len(pcm)) + pcm)
    r = subprocess.run([cli, "1", "64", "512", "2", str(wav), str(w)], capture_output=True, env=env)
    assert r.returncode != 0 and b"only 44.1 kHz" in r.stderr
    r = subprocess.run([cli, "1", "64", "512", "2", str(wav)], capture_output=True, env=env)
    assert r.returncode != 0 and b"no weights" in r.stderr
    # beyond 4 GiB: RF64 containers and streamed data chunks of unknown length fail with a message, not with a wrapped 32-bit size
    rf = tmp_path / "big.wav"
    rf.write_bytes(b"RF64" + struct.pack("<I", 0xFFFFFFFF) + b"WAVEds64" + bytes(64))
    r = subprocess.run([cli, "1", "64", "512", "2", str(rf), str(w)], capture_output=True, env=env)
    assert r.returncode != 0 and b"RF64" in r.stderr
    st = tmp_path / "stream.wav"
    st.write_bytes(b"RIFF" + struct.pack("<I", 0xFFFFFFFF) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 2, 44100, 44100 * 4, 4, 16)
                   + b"data" + struct.pack("<I", 0xFFFFFFFF) + pcm)
    r = subprocess.run([cli, "1", "64", "512", "2", str(st), str(w)], capture_output=True, env=env)
    assert r.returncode != 0 and b"unknown length" in r.stderr


def test_tuning_build_still_compiles():
    """The SRT_TUNING=1 build (alternative tile shapes + the ablation kernels quoted in DESIGN.md §6) is not part of the
    shipped library; keep it from rotting with a syntax-only compile of the three kernel files."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "spleeterrt_amd", "csrc")
    for f in ("srt_nn.hip", "srt_nn2.hip", "srt_nn3.hip", "srt_nn4.hip"):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-std=c++17", "-DSRT_TUNING", "-fsyntax-only", "-Wno-pass-failed",
                            "-I" + os.path.join(ROOT, "include"), "-I" + csrc, os.path.join(csrc, f)], capture_output=True)
        assert r.returncode == 0, r.stderr.decode()[-2000:]


def test_bench_takes_kernel_names_from_the_engine(lib):
    """bench.py no longer keeps a layer -> kernel-symbol table: the engine reports the symbol of every timed launch
    (srtGetTimingKernels, GPU-tested in tests/test_gpu_parity.py).  What can still go stale is the committed PMC summary bench.py
    reads `roofline.traffic` / `mfma_busy_frac` from: every MFMA layer kernel named in the NEWEST summary must be a kernel of the built
    library (a template signature that changed without a refreshed profile would silently turn those fields into null)."""
    import json
    import shutil
    import sys
    if not shutil.which("nm"):
        pytest.skip("nm not available")
    sys.path.insert(0, ROOT)
    import bench
    assert not hasattr(bench, "LAYER_SYMBOL")
    out = subprocess.run(["nm", "-C", lib[1]], capture_output=True, text=True).stdout
    have = set()
    for ln in out.splitlines():
        m = re.search(r"(?:void )?(?:__device_stub__)?(srt_\w+(?:<[^(]*>)?)\(", ln)
        if m:
            have.add(m.group(1))
    for prec in ("f32", "f16"):                             # the newest committed counter profile of each mode (f16: BASELINE configs[4]'s five-stem shape)
        newest = next(p for p, _stems in bench.PMC_SUMMARIES[prec] if os.path.exists(p))
        named = [k for k in json.load(open(newest)) if re.match(r"srt_(enc|dec|up6|head)\w*<", k)]
        assert named, newest
        missing = [k for k in named if not any(bench.same_kernel(k, h) for h in have)]
        assert not missing, "%s names kernels the library does not contain (re-run scripts/profile_gpu.sh): %r" % (newest, missing)


def test_bench_roofline_peaks_follow_the_precision():
    """bench.py prices a kernel against the peak of the MFMA it runs on (157.3 TFLOP/s fp32, 2500 dense fp16), counts the two MFMAs per tap
    of the split mode, and knows the algorithmic bytes of every layer at the mode's element sizes (the HBM roofline of the fp16 path)."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    assert bench.mfma_peak("srt_enc_f16<32, 1, 4, 1, 1, true>") == 2500.0 and bench.mfma_peak("srt_dec_f16<32, 2, 4, 1, 1, true>") == 2500.0
    assert bench.mfma_peak("srt_up6_kernel<8, 64, 32, true, 2>") == 2500.0 and bench.mfma_peak("srt_up6_kernel<8, 64, 32, false, 2>") == 157.3
    assert bench.mfma_peak("srt_dec_wino32<2, 16, 0, 3, 2, 1, 1, 0, 1, 1, 1>") == 157.3 and bench.mfma_peak("srt_enc_mfma2<64, 2, 32, 2, 4, 1, 2, true, 0, false, false>") == 157.3
    assert bench.executed_fraction("srt_enc_f16<32, 1, 4, 1, 1, true>", "f16x2") == 2.0 and bench.executed_fraction("srt_enc_f16<32, 1, 4, 1, 1, true>", "f16") == 1.0
    assert bench.executed_fraction("srt_enc_wino32<2, 16, 1, 0>") == 0.49
    assert bench.executed_fraction("srt_dec_c8<32, 8, 1, 3, true, 0>") == 1.2 and bench.executed_fraction("srt_dec_c8<32, 8, 1, 3, false, 0>") == 1.0   # class-stacked up5: 15 x 32 rows for 25 x 16
    # bytes: up2 at 256 x 1024 reads 512 channels of 8 x 32 and writes 128 of 16 x 64, fp32 / fp16 storage
    assert bench.layer_bytes("up2", "f32", False) == 512 * 8 * 32 * 4 + 128 * 16 * 64 * 4
    assert bench.layer_bytes("up2", "f16", True) == (512 * 8 * 32 + 128 * 16 * 64) * 2
    assert bench.layer_bytes("down2", "f16", True) == 16 * 128 * 512 * 2 + 32 * 64 * 256 * 2 * 2          # raw + act copy, halves
    assert bench.layer_bytes("up6", "f16", True) == 32 * 128 * 512 * 2 + 256 * 1024 * 4                    # up6's output stays fp32


def test_down1_stream_store_count_matches_its_vmcnt_wait(tmp_path):
    """srt_down1_stream_kernel proves "my LDS-DMA pieces have landed" with s_waitcnt vmcnt(nst), nst = the stores a wave issues per interval
    (8 per live 16-row stem group, x2 with fp16 storage).  That holds only if the compiler emits exactly ONE VM instruction per float4 / h4 store and spills
    nothing to scratch.  Checked on the ISA of the six shipped instantiations (four-wave and two-wave form; fp32, planar fp16 and C8 fp16 outputs): store count, scratch size, and the
    vmcnt immediates of the kernel."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "spleeterrt_amd", "csrc")
    asm = tmp_path / "nn2.s"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed", "-Wno-unused-command-line-argument", "--cuda-device-only", "-S",
                        "-I" + os.path.join(ROOT, "include"), "-I" + csrc, os.path.join(csrc, "srt_nn2.hip"), "-o", str(asm)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    txt = asm.read_text().split("\n")
    for mangled, stores_per_group in (("_Z23srt_down1_stream_kernelILi0ELb0ELi4ELb0EEv13SrtConvParams", 8), ("_Z23srt_down1_stream_kernelILi0ELb1ELi4ELb0EEv13SrtConvParams", 16),
                                      ("_Z23srt_down1_stream_kernelILi0ELb0ELi2ELb0EEv13SrtConvParams", 8), ("_Z23srt_down1_stream_kernelILi0ELb1ELi2ELb0EEv13SrtConvParams", 16),   # NW = 2: the two-wave form (one M tile)
                                      ("_Z23srt_down1_stream_kernelILi0ELb1ELi4ELb1EEv13SrtConvParams", 8), ("_Z23srt_down1_stream_kernelILi0ELb1ELi2ELb1EEv13SrtConvParams", 8)):    # C8 outputs: 16-byte stores, 8 per stem
        start = next(i for i, l in enumerate(txt) if l.startswith(mangled + ":"))
        end = next(i for i in range(start, len(txt)) if txt[i].startswith("\t.end_amdhsa_kernel"))
        body = [l.strip() for l in txt[start:end] if l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;")]
        meta = [l for l in txt[start:end] if ".amdhsa_private_segment_fixed_size" in l]
        assert meta and meta[0].split()[-1] == "0", meta                                   # nothing spilled: no hidden scratch stores in vmcnt
        assert not [l for l in body if l.startswith("scratch_")]
        stores = [l for l in body if l.startswith("global_store") or l.startswith("buffer_store")]
        # every store of the kernel sits in the interval loop: two 16-row stem groups (registers 0..7 / 8..15), each behind its own wave-uniform guard
        assert len(stores) == 2 * stores_per_group, (mangled, len(stores))
        want_op = "global_store_dwordx4" if stores_per_group == 8 else "global_store_dwordx2"
        assert all(l.split()[0] == want_op for l in stores), sorted(set(l.split()[0] for l in stores))
        waits = sorted(set(int(m) for l in body if l.startswith("s_waitcnt") for m in re.findall(r"vmcnt\((\d+)\)", l)))
        # the counted waits the source asks for at the top of an interval: both groups live / one group live (the compiler's own waits are vmcnt(0) or small)
        assert 2 * stores_per_group in waits and stores_per_group in waits, (mangled, waits)
        dma = [l for l in body if l.startswith("buffer_load_dwordx4") and " lds" in l]
        assert len(dma) >= 3                                                               # the LDS-DMA pieces the wait is about


def test_down1_f16_store_count_matches_its_vmcnt_wait(tmp_path):
    """srt_down1_f16_kernel (down1 of the fp16 mode on the fp16 MFMA, csrc/srt_nn2.hip) uses the streamed kernel's counted wait: s_waitcnt vmcnt(4 x stems) at the top of an
    interval = "my DMA pieces have landed, the interval's stores may still fly".  Pinned on the ISA of the three instantiations (one, two, three M tiles): exactly two 16-byte
    stores (raw, act) per (stem, channel group) = 8 per M tile, nothing in scratch, every counted immediate 4 .. 24 present, the LDS-DMA pieces and 10 fp16 MFMAs per M tile."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "spleeterrt_amd", "csrc")
    asm = tmp_path / "nn2.s"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed", "-Wno-unused-command-line-argument", "--cuda-device-only", "-S",
                        "-I" + os.path.join(ROOT, "include"), "-I" + csrc, os.path.join(csrc, "srt_nn2.hip"), "-o", str(asm)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    txt = asm.read_text().split("\n")
    for mt in (1, 2, 3):
        mangled = "_Z20srt_down1_f16_kernelILi%dEEv13SrtConvParams" % mt
        start = next(i for i, l in enumerate(txt) if l.startswith(mangled + ":"))
        end = next(i for i in range(start, len(txt)) if txt[i].startswith("\t.end_amdhsa_kernel"))
        body = [l.strip() for l in txt[start:end] if l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;")]
        meta = [l for l in txt[start:end] if ".amdhsa_private_segment_fixed_size" in l]
        assert meta and meta[0].split()[-1] == "0", meta
        assert not [l for l in body if l.startswith("scratch_")]
        stores = [l.split()[0] for l in body if l.startswith("global_store") or l.startswith("buffer_store")]
        assert stores == ["global_store_dwordx4"] * (8 * mt), (mt, stores)
        waits = set(int(m) for l in body if l.startswith("s_waitcnt") for m in re.findall(r"vmcnt\((\d+)\)", l))
        assert {4, 8, 12, 16, 20, 24} <= waits, (mt, sorted(waits))
        assert len([l for l in body if l.startswith("buffer_load_dwordx4") and " lds" in l]) >= 3
        assert len([l for l in body if l.startswith("v_mfma_f32_32x32x16_f16")]) == 10 * mt


def test_c8_kernels_store_count_matches_their_vmcnt_wait(tmp_path):
    """The C8-form fp16 kernels (csrc/srt_nn5.hip) prove "my LDS-DMA pieces of this step have landed" with s_waitcnt vmcnt(pieces still allowed in flight + NST), NST =
    the store instructions of the epilogue a wave issued behind the previous step's DMA (vmcnt retires in issue order).  An NST above the real count would let
    a step read a stage before its DMA landed.  Pinned on the ISA of every shipped instantiation: stores per epilogue copy (the epilogue is emitted twice: in the K-step
    loop and for the workgroup's last unit), their width, no scratch, and the counted wait immediates.  (LW = 1 instantiations: the DMA is issued by loader-only waves
    that never store, so only the ring-depth wait exists there.)"""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "spleeterrt_amd", "csrc")
    asm = tmp_path / "nn5.s"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed", "-Wno-unused-command-line-argument", "--cuda-device-only", "-S",
                        "-I" + os.path.join(ROOT, "include"), "-I" + csrc, os.path.join(csrc, "srt_nn5.hip"), "-o", str(asm)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    txt = asm.read_text().split("\n")
    kernels = [m.group(1) for m in (re.match(r"(_Z10srt_(?:enc|dec)_c8I\w+):", l) for l in txt) if m]
    assert len(kernels) == 7, kernels                       # 2 tile shapes x (encoder, decoder, class-stacked decoder) + the decoder with two sub-tiles per wave; the LW = 1 forms exist in the tuning library only
    for mangled in kernels:
        start = next(i for i, l in enumerate(txt) if l.startswith(mangled + ":"))
        end = next(i for i in range(start, len(txt)) if txt[i].startswith("\t.end_amdhsa_kernel"))
        body = [l.strip() for l in txt[start:end] if l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;")]
        meta = [l for l in txt[start:end] if ".amdhsa_private_segment_fixed_size" in l]
        assert meta and meta[0].split()[-1] == "0", (mangled, meta)
        assert not [l for l in body if l.startswith("scratch_")]
        stores = [l.split()[0] for l in body if l.startswith("global_store") or l.startswith("buffer_store")]
        waits = sorted(set(int(m) for l in body if l.startswith("s_waitcnt") for m in re.findall(r"vmcnt\((\d+)\)", l)))
        targs = [int(x) for x in re.findall(r"L[ib](\d+)E", mangled)]
        enc = "srt_enc_c8" in mangled
        sw, lw = targs[0], targs[3] if enc else targs[5]
        nrw = 1 if enc or len(targs) < 9 else targs[8]      # decoder: sub-tiles per wave of the LW = 0 form (NRW)
        nr = 2 if lw else nrw                               # sub-tiles (hence epilogue copies of the stores) per computing wave
        if enc:                                             # raw + act: 2 + 2 sixteen-byte stores per sub-tile and epilogue
            assert stores == ["global_store_dwordx4"] * (8 * nr), (mangled, stores)
            if not lw:                                      # LW = 0: the wait allows the 4 (2 without the act copy: down6) stores behind the DMA
                assert 4 in waits and 2 in waits and 0 in waits, (mangled, waits)
        else:
            cs = targs[4] == 1                              # class-stacked (up5, Cout = 16): two channel groups x two rows of 16-byte stores per epilogue, 15 weight rows per chunk
            nst = 4 if cs else 8
            assert stores == ["global_store_dwordx4"] * (2 * nst * nr), (mangled, stores)
            nlw = 4 if lw else 8
            nsy, ni = targs[1], targs[2]
            th, tw = nsy * (32 // sw), sw
            npp = -(-(2 * ni * (th + 2) * (tw + 2)) // 64)  # patch pieces of 1 KiB per stage (two k-groups of 16-byte pixel slots)
            assert npp == {(32, 8): 11, (16, 2): 14, (32, 16): 20}[(sw, nsy)]
            dpw = -(-npp // nlw) + -(-(15 if cs else 25) // nlw)     # DMA instructions per loader wave and K step
            assert dpw in waits, (mangled, dpw, waits)      # ring depth 3: one step's pieces may stay in flight
            if not lw:
                assert dpw + nst in waits, (mangled, waits) # ... + the NST epilogue stores of one sub-tile issued behind them
                if nr == 2:
                    assert dpw + 2 * nst in waits, (mangled, waits)    # ... or of both
        assert len([l for l in body if l.startswith("buffer_load_dwordx4") and " lds" in l]) >= 2
