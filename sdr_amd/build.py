"""Build libsdrx.so (HIP, gfx950) in-tree.  hipcc cross-compiles without a GPU.

    python -m sdr_amd.build [--force]

-ffp-contract=off is part of the numerics contract: the kernels must round after every
multiply and every add, like the reference's CPU loops.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsdrx.so")
SOURCES = ["api.hip", "fir_kernels.hip", "demod_kernels.hip", "psk_kernels.hip"]
HEADERS = [os.path.join(CSRC, "sdrx_internal.h"), os.path.join(HERE, "..", "include", "sdrx.h"),
           os.path.join(HERE, "..", "include", "sdrx_math.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-Wno-comment"]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
