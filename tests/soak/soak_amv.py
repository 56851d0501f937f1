"""One-off soak of the verified AM chain (config 1): random filters, lengths, channel counts, sample kinds (8-bit integers,
fractions, a mix by tile, wide dynamic range, thresholds scaled onto the envelope), rail layouts and call splits, every form
(SDRX_FIR_AMV 1 default, 2 packed fma, 3 fp32 matrix, 4 int8 matrix) against the oracle's chain, bit for bit.
usage: python tests/soak/soak_amv.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import torch
import _oracle as O
from sdr_amd import _lib, device

device.init(0)
L, Lx = O.lib(), _lib.lib()
t_end = time.time() + float(sys.argv[1] if len(sys.argv) > 1 else 60)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)


def oracle_bits(taps, re, im, hI, hQ):
    bits = np.zeros(re.size + 1, dtype=np.uint8)
    m = L.orc_chain_fir_am(taps, taps.size, hI, hQ, np.ascontiguousarray(re), np.ascontiguousarray(im), re.size, 1, bits)
    return bits[:m]


rounds = outputs = 0
while time.time() < t_end:
    rounds += 1
    T = int(rng.choice([1, 2, 3, 7, 15, 16, 17, 31, 48, 49, 50, 63, 64, 65, 100, 113, 114, 129, 200, 241, 242, 300, 700]))
    nchan = int(rng.choice([1, 1, 2, 3, 5]))
    n = int(rng.choice([1, 37, 2047, 2048, 2049, 4096, 5000, 12345, 40000, 70001]))
    taps = rng.normal(0, 1, T) * np.exp(rng.uniform(-rng.choice([0.0, 3.0, 12.0]), 0, T))
    kind = rng.choice(["int", "int", "frac", "mixed", "wide", "bytes", "bytes"])
    raw = None
    if kind == "bytes":  # the radio's raw {I, Q} bytes (sample = byte - 127: every byte value, 255 = +128 included)
        n = int(rng.choice([2048, 4096, 4096 + 8, 40000, 70001, 16 * 2048, 16 * 2048 + 1000]))
        raw = rng.integers(0, 256, (nchan, n, 2)).astype(np.uint8)
        if rng.random() < 0.5:
            raw = (127 + np.clip(np.round(rng.normal(0, 40, (nchan, n, 2))), -127, 128)).astype(np.uint8)
        re, im = raw[:, :, 0].astype(np.float64) - 127.0, raw[:, :, 1].astype(np.float64) - 127.0
    elif kind in ("int", "mixed"):
        re = rng.integers(-127, 128, (nchan, n)).astype(np.float64)
        im = rng.integers(-127, 128, (nchan, n)).astype(np.float64)
        if kind == "mixed":
            for _ in range(int(rng.integers(1, 6))):
                c, p = int(rng.integers(0, nchan)), int(rng.integers(0, n))
                (re if rng.random() < 0.5 else im)[c, p] = rng.choice([0.5, 128.0, -128.0, 127.5, 1e6, np.nan, np.inf, -300.25])
    elif kind == "frac":
        re, im = rng.normal(0, 60, (nchan, n)), rng.normal(0, 60, (nchan, n))
    else:
        re = rng.normal(0, 1, (nchan, n)) * np.exp(rng.uniform(-20, 8, (nchan, n)))
        im = rng.normal(0, 1, (nchan, n)) * np.exp(rng.uniform(-20, 8, (nchan, n)))
    with np.errstate(all="ignore"):
        env = np.abs(np.convolve(np.nan_to_num(re[0], posinf=0, neginf=0) + 1j * np.nan_to_num(im[0], posinf=0, neginf=0), taps)[:n])
        med = np.median(env[min(T, n - 1):]) if n > 1 else 1.0
    if med > 0 and np.isfinite(med):
        taps = taps * (80.0 / med)  # the envelope straddles the threshold
    cuts = [0] + sorted(set(int(v) for v in rng.integers(1, max(n, 2), int(rng.integers(0, 3))) if v < n)) + [n]
    if raw is not None and rng.random() < 0.7:  # rows on 16-byte boundaries: the multi-tile byte kernel takes the inner tiles
        cuts = sorted(set([0, n] + [int(v) // 8 * 8 for v in cuts[1:-1]]))
    hI = [np.zeros(max(T - 1, 1)) for _ in range(nchan)]
    hQ = [np.zeros(max(T - 1, 1)) for _ in range(nchan)]
    with np.errstate(all="ignore"):
        want = np.stack([np.concatenate([oracle_bits(taps, re[c, a:b], im[c, a:b], hI[c], hQ[c]) for a, b in zip(cuts[:-1], cuts[1:])])
                         for c in range(nchan)])
    split = bool(rng.random() < 0.3) and raw is None
    d_re, d_im = torch.tensor(re, device="cuda"), torch.tensor(im, device="cuda")
    d_raw = torch.tensor(raw, device="cuda") if raw is not None else None
    for form in (1, 2, 3, 4):
        Lx.sdrx_tune(b"SDRX_FIR_AMV", str(form).encode())
        Lx.sdrx_tune(b"SDRX_AMV_NB", b"4" if rng.random() < 0.3 else b"2")
        ch = device.FirChain(taps, nchan=nchan, decim=1)
        got = torch.zeros(nchan, n, dtype=torch.uint8, device="cuda")
        for a, b in zip(cuts[:-1], cuts[1:]):
            if d_raw is not None:
                x = d_raw[:, a:b].contiguous()
            else:
                x = (d_re[:, a:b].contiguous(), d_im[:, a:b].contiguous()) if split else torch.stack([d_re[:, a:b], d_im[:, a:b]], dim=-1).contiguous()
            o = torch.zeros(nchan, b - a, dtype=torch.uint8, device="cuda")
            ch.fir_am(x, o)
            got[:, a:b] = o
        torch.cuda.synchronize()
        g = got.cpu().numpy()
        if not np.array_equal(g, want):
            bad = np.argwhere(g != want)
            print("MISMATCH", dict(T=T, nchan=nchan, n=n, kind=str(kind), cuts=cuts, split=split, form=form), bad[:5], flush=True)
            sys.exit(1)
        outputs += nchan * n
    if rounds % 20 == 0:
        print("round", rounds, "outputs", outputs, "ones", float(want.mean()), flush=True)
print("soak_amv: %d rounds, %d outputs x 4 forms, all bit-identical" % (rounds, outputs // 4))
