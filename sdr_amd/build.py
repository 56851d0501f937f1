"""Build libsdrx.so (HIP, gfx950) in-tree.  hipcc cross-compiles without a GPU.

    python -m sdr_amd.build [--force]

-ffp-contract=off is part of the numerics contract: the kernels must round after every
multiply and every add, like the reference's CPU loops.  Each source is compiled
to its own object (in parallel, only when stale), then linked.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "..", "build", "obj")
LIB = os.path.join(HERE, "libsdrx.so")
SOURCES = ["api.hip", "fir_kernels.hip", "fir_kernels_fma.hip", "fir_kernels_f32.hip", "fir_mfma.hip", "demod_kernels.hip", "mod_kernels.hip", "psk_kernels.hip", "framer_kernels.hip", "dsp_kernels.hip"]
HEADERS = [os.path.join(CSRC, h) for h in ("sdrx_internal.h", "fir_common.h", "fir_impl.inc")] + [
    os.path.join(HERE, "..", "include", "sdrx.h"), os.path.join(HERE, "..", "include", "sdrx_math.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wno-comment"]
# Per-source additions (none in use).  Tried for psk_kernels.hip: -mllvm -amdgpu-sched-strategy=max-ilp -- the PSK loops run
# ONE wave per SIMD, which pays 8.4 cycles for an instruction that waits for the one before it and 5.2 for an independent one
# (profiles/probes/issue_probe.hip) -- but the strategy leaves the loop blocks' order as it was (39 % of their vector
# instructions still follow their producer directly) and the loops' time unchanged (2.742 against 2.749 ms, 1024 x 2^16).
EXTRA_FLAGS = {}


def _newer(path, deps):
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    return any(os.path.getmtime(d) > t for d in deps)


def stale():
    return _newer(LIB, [os.path.join(CSRC, s) for s in SOURCES] + HEADERS)


def build(force=False, verbose=True):
    if not force and not stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s + ".o")
        if force or _newer(obj, [src, os.path.abspath(__file__)] + HEADERS):
            jobs.append([hipcc] + FLAGS + EXTRA_FLAGS.get(s, []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(6, int(os.environ.get("MAX_JOBS", "6")))) as ex:
        list(ex.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [os.path.join(OBJ, s + ".o") for s in SOURCES])
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
