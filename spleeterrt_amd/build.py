"""Build libspleeterrt_amd.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m spleeterrt_amd.build [--force] [--tuning]

--tuning builds a SECOND library, libspleeterrt_amd_tuning.so, with -DSRT_TUNING: the alternative tile shapes, kernel
variants and ablation builds measured for DESIGN.md (selected at run time with SRT_TUNE=key=value,...).  It is loaded only
when SPLEETERRT_LIB points at it (scripts/gpu_tune.sh); the product library never contains those variants.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so travels to the GPU box.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
INC = os.path.join(os.path.dirname(PKG), "include")
SO = os.path.join(PKG, "libspleeterrt_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# SRT_TUNING=1 adds the alternative tile shapes and the ablation builds (selected at run time with SRT_TUNE=...; see
# csrc/srt_nn2.hip); a default build holds only the shipped configuration.
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + INC, "-I" + CSRC, "-Wno-unused-result", "-Wno-unused-value", "-Wno-pass-failed", "-fvisibility=hidden"]
SO_TUNING = os.path.join(PKG, "libspleeterrt_amd_tuning.so")
# per-file flags.  srt_nn4.hip: no SLP vectorisation - packed f32 adds (v_pk_add_f32) beside MFMAs cost more than the two scalar adds
# they replace (MI355X_MICROARCH.md, "price of one filler beside MFMAs"), and its transform arithmetic is issued in the MFMAs' shadow.
FILE_FLAGS = {"srt_nn4.hip": ["-fno-slp-vectorize"]}
for _kv in os.environ.get("SRT_FILE_FLAGS", "").split(";"):                # measurement aid: "srt_nn2.hip=-fno-slp-vectorize;srt_nn4.hip=-O3"
    if "=" in _kv:
        FILE_FLAGS[_kv.split("=", 1)[0]] = _kv.split("=", 1)[1].split()


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(out, deps):
    return not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)


def build(force=False, verbose=True, tuning=False):
    tuning = tuning or os.environ.get("SRT_TUNING") == "1"
    flags = (["-DSRT_TUNING"] if tuning else []) + BASE_FLAGS
    bdir = os.path.join(PKG, "build_tuning" if tuning else "build")
    so = SO_TUNING if tuning else SO
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
           [os.path.join(INC, f) for f in os.listdir(INC) if f.endswith(".h")]
    objs, procs = [], []
    os.makedirs(bdir, exist_ok=True)
    tag, tagfile = " ".join(flags), os.path.join(bdir, ".flags")
    if not os.path.exists(tagfile) or open(tagfile).read() != tag:      # flags changed: rebuild all
        force = True
        open(tagfile, "w").write(tag)
    for src in sources():
        obj = os.path.join(bdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            if verbose:
                print("[build] hipcc -c", os.path.basename(src), "(tuning)" if tuning else "", flush=True)
            procs.append((src, subprocess.Popen([HIPCC] + flags + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj])))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    if force or procs or _stale(so, objs):
        if verbose:
            print("[build] link", os.path.basename(so), flush=True)
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
    return so


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, tuning="--tuning" in sys.argv))
