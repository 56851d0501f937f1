#!/usr/bin/env python3
"""Extract the inline MATLAB golden vectors held by the reference's own unit tests.

The reference's gtest files (RadioNode/src/test/TEST_*.cpp) pin every stage of
the AM / FM / BPSK / QPSK path with array literals pasted from MATLAB.  This
script pulls those array literals out (data only: names + numbers, no code) and
stores them in ``tests/golden/reference_vectors.npz`` so the oracle and the HIP
path can be checked against them on machines where /root/reference is absent.

Run once in the build container:

    python tests/golden/extract_reference_vectors.py [/root/reference]

Key naming:  <FILE>__<SUITE>__<TEST>__<array>  (globals: <FILE>____<array>)
e.g.  TEST_BPSK__BPSK__decode_pulseShape__RxCorrelated
"""
import hashlib
import json
import os
import re
import sys

import numpy as np

FILES = [
    "TEST_AM.cpp",
    "TEST_FM.cpp",
    "TEST_BPSK.cpp",
    "TEST_QPSK.cpp",
    "TEST_Modulator_Demodulator.cpp",
    "TEST_Resample.cpp",
]

_ARRAY_RE = re.compile(
    r"(?:static\s+)?(?:const\s+)?(uint8_t|RADIO_DATA_TYPE|double|int)\s+(\w+)\s*\[[^\]]*\]\s*=\s*\{([^}]*)\}\s*;",
    re.S,
)
_TEST_RE = re.compile(r"\bTEST\s*\(\s*(\w+)\s*,\s*(\w+)\s*\)")


def strip_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return text


def parse_file(path: str):
    with open(path, "r", errors="replace") as f:
        text = strip_comments(f.read())
    base = os.path.splitext(os.path.basename(path))[0]
    tests = [(m.start(), m.group(1), m.group(2)) for m in _TEST_RE.finditer(text)]
    out = {}
    for m in _ARRAY_RE.finditer(text):
        ctype, name, body = m.group(1), m.group(2), m.group(3)
        suite, test = "", ""
        for pos, s, t in tests:
            if pos < m.start():
                suite, test = s, t
        toks = [t.strip() for t in body.replace("\n", " ").split(",")]
        toks = [t for t in toks if t]
        if not toks:
            continue
        if ctype == "uint8_t":
            arr = np.array([int(t) for t in toks], dtype=np.uint8)
        else:
            arr = np.array([float(t) for t in toks], dtype=np.float64)
        key = f"{base}__{suite}__{test}__{name}"
        if key in out:
            raise RuntimeError(f"duplicate key {key}")
        out[key] = arr
    return out


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    test_dir = os.path.join(ref, "RadioNode", "src", "test")
    here = os.path.dirname(os.path.abspath(__file__))
    allv = {}
    for fn in FILES:
        allv.update(parse_file(os.path.join(test_dir, fn)))
    npz = os.path.join(here, "reference_vectors.npz")
    np.savez_compressed(npz, **allv)
    manifest = {
        k: {
            "dtype": str(v.dtype),
            "len": int(v.size),
            "sha256": hashlib.sha256(v.tobytes()).hexdigest(),
        }
        for k, v in sorted(allv.items())
    }
    with open(os.path.join(here, "reference_vectors.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print(f"{len(allv)} arrays -> {npz} ({os.path.getsize(npz)} bytes)")
    for k, v in sorted(allv.items()):
        print(f"  {k}: {v.dtype}[{v.size}]")


if __name__ == "__main__":
    main()
