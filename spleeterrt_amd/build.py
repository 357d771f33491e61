"""Build libspleeterrt_amd.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m spleeterrt_amd.build [--force]

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
FLAGS = (["-DSRT_TUNING"] if os.environ.get("SRT_TUNING") == "1" else []) + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + INC, "-I" + CSRC, "-Wno-unused-result", "-Wno-unused-value", "-Wno-pass-failed", "-fvisibility=hidden"]
FLAGS_TAG = os.path.join(PKG, "build", ".flags")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(out, deps):
    return not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)


def build(force=False, verbose=True):
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
           [os.path.join(INC, f) for f in os.listdir(INC) if f.endswith(".h")]
    objs, procs = [], []
    os.makedirs(os.path.join(PKG, "build"), exist_ok=True)
    tag = " ".join(FLAGS)
    if not os.path.exists(FLAGS_TAG) or open(FLAGS_TAG).read() != tag:      # flags changed (e.g. SRT_TUNING toggled): rebuild all
        force = True
        open(FLAGS_TAG, "w").write(tag)
    for src in sources():
        obj = os.path.join(PKG, "build", os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            if verbose:
                print("[build] hipcc -c", os.path.basename(src), flush=True)
            procs.append((src, subprocess.Popen([HIPCC] + FLAGS + ["-c", src, "-o", obj])))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    if force or procs or _stale(SO, objs):
        if verbose:
            print("[build] link", os.path.basename(SO), flush=True)
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
