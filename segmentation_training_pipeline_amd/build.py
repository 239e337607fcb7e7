"""Builds libstp_hip.so (the C-ABI shared library, include/stp_hip.h) for gfx950 with hipcc, and libstp_hip_f16.so - the same
sources with -DSTP_STORAGE_F16=1: IEEE-half storage and v_mfma_*_f16 instead of bfloat16 (same symbols, STP_F16 dtype code).

In-tree build: the .so lands next to this file so it travels with the repo snapshot to the
GPU box.  hipcc cross-compiles without a GPU.  Usage: ``python -m segmentation_training_pipeline_amd.build``.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libstp_hip.so")
LIB_F16 = os.path.join(HERE, "libstp_hip_f16.so")
OBJ = os.path.join(HERE, "csrc", "_obj")
# (library, object suffix, extra flags): the 16-bit storage format is a build parameter of the kernel set (csrc/common.h)
VARIANTS = [(LIB, ".o", []), (LIB_F16, ".f16.o", ["-DSTP_STORAGE_F16=1"])]
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wno-unused-result"]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libstp_hip.so")
    return exe


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "stp_hip.h"))
    jobs = []
    link = []
    for lib, suffix, extra in VARIANTS:
        objs = []
        stale = False
        for src in sources():
            s = os.path.join(CSRC, src)
            o = os.path.join(OBJ, src[:-4] + suffix)
            objs.append(o)
            if force or _stale(o, [s] + headers):
                jobs.append([hipcc] + FLAGS + extra + ["-c", s, "-o", o])
                stale = True
        link.append((lib, objs, stale))

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(run, jobs))
    for lib, objs, stale in link:
        if stale or force or _stale(lib, objs):
            run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
