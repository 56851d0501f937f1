#!/usr/bin/env python3
"""Achieved HBM rate of the small streaming kernels of the path (not the bench line): AM envelope
threshold, ones/zeros count, FSK correlator, PSK ambiguity pass, rtl_fm int16 chain.  Inputs resident
in HBM, 20 timed launches after 5 warm-up, torch events on the launch stream.
    python profiles/kernel_rates.py > profiles/rNN_kernel_rates.json   (on the GPU box)"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sdr_amd import _lib, device  # noqa: E402

device.init(0)
dev = torch.device("cuda", 0)
PEAK = 8000.0


def timed(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters / 1e3


rows = []
n = 1 << 28
re = torch.randint(-127, 128, (n,), device=dev).to(torch.float64)
im = torch.randint(-127, 128, (n,), device=dev).to(torch.float64)
for ds in (1, 8):
    out = torch.zeros((n + ds - 1) // ds, dtype=torch.uint8, device=dev)
    t = timed(lambda: device.am_demod(re, im, out, ds))
    alg = (16.0 / ds if ds > 1 else 16.0) * n + out.numel()  # ds > 1 touches one sample in ds (whole 64 B lines move)
    rows.append({"kernel": f"am_demod ds={ds}", "samples": n, "ms": t * 1e3, "Msamples_per_s": n / t / 1e6,
                 "algorithmic_GBps": alg / t / 1e9, "frac_of_8TBps": alg / t / 1e9 / PEAK})
cnt = torch.zeros(1, dtype=torch.int32, device=dev)
t = timed(lambda: device.ones_zeros(re, im, cnt))
rows.append({"kernel": "ones_zeros", "samples": n, "ms": t * 1e3, "Msamples_per_s": n / t / 1e6,
             "algorithmic_GBps": 16.0 * n / t / 1e9, "frac_of_8TBps": 16.0 * n / t / 1e9 / PEAK})
for ds in (100, 2000):
    w1 = torch.randn(ds, dtype=torch.float64, device=dev)
    w2 = torch.randn(ds, dtype=torch.float64, device=dev)
    out = torch.zeros(n // ds, dtype=torch.uint8, device=dev)
    t = timed(lambda: device.fm_demod(re, w1, w2, out, ds))
    rows.append({"kernel": f"fsk_corr ds={ds}", "samples": n, "ms": t * 1e3, "Msamples_per_s": n / t / 1e6,
                 "algorithmic_GBps": 8.0 * n / t / 1e9, "frac_of_8TBps": 8.0 * n / t / 1e9 / PEAK})
del re, im
nchan, nb = 1024, 1 << 18
raw = torch.randint(0, 256, (nchan, nb, 2), device=dev, dtype=torch.uint8)
fm = device.RtlFm(nchan, 10)
pcm = torch.zeros(nchan, nb // 10 + 2, dtype=torch.int16, device=dev)
t = timed(lambda: fm.demod(raw, pcm))
tot = nchan * nb
rows.append({"kernel": "rtlfm int16 chain ds=10", "samples": tot, "ms": t * 1e3, "Msamples_per_s": tot / t / 1e6,
             "algorithmic_GBps": 2.2 * tot / t / 1e9, "frac_of_8TBps": 2.2 * tot / t / 1e9 / PEAK})
del raw, pcm
# bit-domain back end: 1024 channels x 2^18 bits, (a) dense framed packets, (b) noise with no header
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from signals import framed_bit_stream  # noqa: E402

nb = 1 << 18
one = framed_bit_stream(nb, 1, packet_every=64, p_bad_crc=0.05, p_bad_len=0.05)
dense = torch.tensor(np.stack([np.roll(one, 37 * c) for c in range(64)]), device=dev).repeat(16, 1).contiguous()
noise = torch.randint(0, 2, (1024, nb), device=dev, dtype=torch.uint8)
ev = torch.zeros(1024, 256, 16, dtype=torch.uint8, device=dev)
pay = torch.zeros(1024, 256, 250, dtype=torch.uint8, device=dev)
summ = torch.zeros(1024, 4, dtype=torch.int64, device=dev)
L = _lib.lib()
import ctypes as C  # noqa: E402
for name, bits in (("deframe dense packets", dense), ("deframe noise", noise)):
    def run():
        _lib.check(L.sdrx_deframe_dev(C.c_void_p(bits.data_ptr()), nb, 1024, nb, 1, C.c_void_p(ev.data_ptr()),
                                      C.c_void_p(pay.data_ptr()), 256, C.c_void_p(summ.data_ptr()), None))
    torch.cuda.synchronize()
    t = timed(run, iters=10, warm=2)
    tot = 1024 * nb
    rows.append({"kernel": name + " (1024 ch x 2^18 bits)", "samples": tot, "ms": t * 1e3, "Msamples_per_s": tot / t / 1e6,
                 "algorithmic_GBps": 1.0 * tot / t / 1e9, "frac_of_8TBps": 1.0 * tot / t / 1e9 / PEAK,
                 "events_per_channel": float(summ[:, 0].double().mean().item())})
print(json.dumps({"peak_GBps": PEAK, "rows": rows}, indent=1))
