"""Build libfaceformer_hip.so (gfx950) in-tree with hipcc.  No torch involvement: the library is a
plain C-ABI shared object (see include/faceformer_hip.h) linked only against the HIP runtime.

    python -m faceformer_amd.hip.build [--force] [--verbose]
"""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB_PATH = os.path.join(HERE, "libfaceformer_hip.so")
BUILD_DIR = os.path.join(HERE, "build")
ARCH = "gfx950"
SOURCES = ["ff_rowops.hip", "ff_gemm.hip", "ff_gemm_x3.hip", "ff_attention.hip", "ff_attention_x2h.hip", "ff_attention_general.hip", "ff_pointer.hip", "ff_engine.hip"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-I" + INCLUDE, "-I" + CSRC,
         "-Wall", "-Wno-unused-function"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built on this machine")
    return exe


def _digest(paths, flags=None):
    h = hashlib.sha256()
    h.update(" ".join(FLAGS if flags is None else flags).encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link the shared library; returns its path."""
    return _build(BUILD_DIR, LIB_PATH, FLAGS, force, verbose)


def _build(BUILD_DIR, LIB_PATH, FLAGS, force, verbose):
    os.makedirs(BUILD_DIR, exist_ok=True)
    headers = [os.path.join(INCLUDE, "faceformer_hip.h"), os.path.join(CSRC, "ff_common.h"), os.path.join(CSRC, "ff_device.h")]
    hipcc = _hipcc()
    objs = []
    relink = force or not os.path.exists(LIB_PATH)
    for src in SOURCES:
        spath = os.path.join(CSRC, src)
        obj = os.path.join(BUILD_DIR, src.replace(".hip", ".o"))
        stamp = obj + ".sha"
        dig = _digest([spath] + headers, FLAGS)
        old = open(stamp).read() if os.path.exists(stamp) else ""
        if force or not os.path.exists(obj) or old != dig:
            cmd = [hipcc] + FLAGS + ["-c", spath, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
            with open(stamp, "w") as f:
                f.write(dig)
            relink = True
        objs.append(obj)
    if relink:
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv)
    print(path)
