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
torch.manual_seed(0)
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
cc = torch.full((512,), 250000, dtype=torch.int32, device=dev)
t = timed(lambda: device.ones_zeros_confidence(re.view(512, -1), im.view(512, -1), 500000, cc))
rows.append({"kernel": "ones_zeros confidence walk (512 ch x 2^19)", "samples": n, "ms": t * 1e3, "Msamples_per_s": n / t / 1e6,
             "algorithmic_GBps": 16.0 * n / t / 1e9, "frac_of_8TBps": 16.0 * n / t / 1e9 / PEAK})
cc1 = torch.full((1,), 250000, dtype=torch.int32, device=dev)
n1 = 1200000  # one radio, the receiver's block (m_SampleRate / 2)
t = timed(lambda: device.ones_zeros_confidence(re[:n1].view(1, -1), im[:n1].view(1, -1), 500000, cc1))
rows.append({"kernel": "ones_zeros confidence walk (1 ch x 1.2M, the receiver's block)", "samples": n1, "ms": t * 1e3,
             "Msamples_per_s": n1 / t / 1e6, "algorithmic_GBps": 16.0 * n1 / t / 1e9, "frac_of_8TBps": 16.0 * n1 / t / 1e9 / PEAK})
for ds in (100, 250, 2000):
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
fmf = device.RtlFmFull(nchan, 10, squelch_level=100, post_downsample=4, deemph_a=3, dc_block=True)
pcmf = torch.zeros(nchan, nb // 10 + 8, dtype=torch.int16, device=dev)
t = timed(lambda: fmf.demod(raw, pcmf))
rows.append({"kernel": "rtlfm full_demod (squelch, /4, deemph, dc block) ds=10", "samples": tot, "ms": t * 1e3,
             "Msamples_per_s": tot / t / 1e6, "algorithmic_GBps": 4.2 * tot / t / 1e9, "frac_of_8TBps": 4.2 * tot / t / 1e9 / PEAK})
del raw, pcm, pcmf
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
b2b = torch.tensor(np.stack([framed_bit_stream(nb, 10 + c, packet_every=1, p_false_uw=0.0) for c in range(64)]), device=dev).repeat(16, 1).contiguous()
for name, bits in (("deframe packets + a bare unique word every ~500 bits", dense), ("deframe back-to-back packets", b2b), ("deframe noise", noise)):
    def run():
        _lib.check(L.sdrx_deframe_dev(C.c_void_p(bits.data_ptr()), nb, 1024, nb, 1, C.c_void_p(ev.data_ptr()),
                                      C.c_void_p(pay.data_ptr()), 256, C.c_void_p(summ.data_ptr()), None))
    torch.cuda.synchronize()
    t = timed(run, iters=10, warm=2)
    tot = 1024 * nb
    rows.append({"kernel": name + " (1024 ch x 2^18 bits)", "samples": tot, "ms": t * 1e3, "Msamples_per_s": tot / t / 1e6,
                 "algorithmic_GBps": 1.0 * tot / t / 1e9, "frac_of_8TBps": 1.0 * tot / t / 1e9 / PEAK,
                 "events_per_channel": float(summ[:, 0].double().mean().item())})
del dense, b2b, noise, ev, pay, summ
# DSP primitives: 16384 rows x 2^13 doubles (1 GiB), read + write = 16 B per sample
nchan, n = 16384, 1 << 13
x = torch.randn(nchan, n, dtype=torch.float64, device=dev) * 50
st = torch.zeros(nchan, 4, dtype=torch.float64, device=dev)
mx = torch.zeros(nchan, dtype=torch.float64, device=dev)
st6 = torch.zeros(nchan, 6, dtype=torch.float64, device=dev)
z = torch.randn(nchan * 2, n // 4, dtype=torch.complex128, device=dev)  # 32768 transforms of 2048 points
coef = device.notch_design(0.9, 0.05)


def _norm():
    mx.zero_()
    device.normalize(x, mx)


for name, fn, bytes_per in (("iir dc_block", lambda: device.iir(device.IIR_DC_BLOCK, 0.995, x), 16.0),
                            ("iir lowpass1", lambda: device.iir(device.IIR_LOWPASS, 0.06, x, state=st), 16.0),
                            ("iir notch", lambda: device.iir(device.IIR_NOTCH, coef, x, state=st), 16.0),
                            ("iir integrate", lambda: device.iir(device.IIR_INTEGRATE, 1e-9, x), 16.0),
                            ("normalize (max pass + scale pass)", _norm, 24.0),
                            ("fifth_order decimate by 2", lambda: device.fifth_order_decimate(x, 2, st6), 4.0 + 4.0),
                            ("fft 2048 points x 32768", lambda: device.fft(z), 16.0)):
    t = timed(fn, iters=10, warm=2)
    tot = nchan * n
    rows.append({"kernel": name + " (16384 rows x 2^13)", "samples": tot, "ms": t * 1e3, "Msamples_per_s": tot / t / 1e6,
                 "algorithmic_GBps": bytes_per * tot / t / 1e9, "frac_of_8TBps": bytes_per * tot / t / 1e9 / PEAK})
del x
# the recurrences on a bank with more, shorter rows (8 waves per SIMD instead of one)
x = torch.randn(131072, 1024, dtype=torch.float64, device=dev) * 50
st2 = torch.zeros(131072, 4, dtype=torch.float64, device=dev)
for name, fn in (("iir dc_block", lambda: device.iir(device.IIR_DC_BLOCK, 0.995, x)),
                 ("iir lowpass1", lambda: device.iir(device.IIR_LOWPASS, 0.06, x, state=st2)),
                 ("iir notch", lambda: device.iir(device.IIR_NOTCH, coef, x, state=st2)),
                 ("iir integrate", lambda: device.iir(device.IIR_INTEGRATE, 1e-9, x))):
    t = timed(fn, iters=10, warm=2)
    tot = x.numel()
    rows.append({"kernel": name + " (131072 rows x 2^10)", "samples": tot, "ms": t * 1e3, "Msamples_per_s": tot / t / 1e6,
                 "algorithmic_GBps": 16.0 * tot / t / 1e9, "frac_of_8TBps": 16.0 * tot / t / 1e9 / PEAK})
del x, st2
for nfft in (64, 256, 1024, 4096):  # the other lengths, same 2^26 points
    z = z.view(-1, nfft)
    t = timed(lambda: device.fft(z), iters=10, warm=2)
    tot = nchan * n
    rows.append({"kernel": f"fft {nfft} points x {z.shape[0]}", "samples": tot, "ms": t * 1e3, "Msamples_per_s": tot / t / 1e6,
                 "algorithmic_GBps": 16.0 * tot / t / 1e9, "frac_of_8TBps": 16.0 * tot / t / 1e9 / PEAK})
del z
# round 2: the modulators as device kernels (output-bound: 8 B written per output sample), the PSK stages of cfg3
nchan, nbit, spb = 1024, 1 << 14, 16
bits = torch.randint(0, 2, (nchan, nbit), device=dev, dtype=torch.uint8)
mod = torch.zeros(nchan, nbit * spb, dtype=torch.float64, device=dev)
t = timed(lambda: device.am_modulate(bits, mod))
tot = nchan * nbit * spb
rows.append({"kernel": "am_modulate (1024 ch x 2^14 bits x 16 samples per bit)", "samples": tot, "ms": t * 1e3, "Msamples_per_s": tot / t / 1e6,
             "algorithmic_GBps": 8.0 * tot / t / 1e9, "frac_of_8TBps": 8.0 * tot / t / 1e9 / PEAK})
wm, wsp = device.fm_tables(2.0e6, 19000.0, 20000.0, spb, True)
wm, wsp = torch.tensor(wm, device=dev), torch.tensor(wsp, device=dev)
t = timed(lambda: device.fm_modulate(bits, wm, wsp, mod))
rows.append({"kernel": "fm_modulate (1024 ch x 2^14 bits x 16 samples per bit)", "samples": tot, "ms": t * 1e3, "Msamples_per_s": tot / t / 1e6,
             "algorithmic_GBps": 8.0 * tot / t / 1e9, "frac_of_8TBps": 8.0 * tot / t / 1e9 / PEAK})
del mod, bits
for kind, tag in ((_lib.QPSK, "QPSK"), (_lib.BPSK, "BPSK")):
    nchan, n = 1 << 14, 1 << 12
    xq = torch.randn(nchan, n, 2, device=dev, dtype=torch.float64) * 0.3
    bank = device.PskBank(kind, nchan, n)
    dec = torch.zeros(nchan, bank.n_dec, dtype=torch.uint8, device=dev)
    m = n // 4
    mf = torch.zeros(nchan, m, 2, dtype=torch.float64, device=dev)
    vp = lambda t_: C.c_void_p(t_.data_ptr())  # noqa: E731
    st_ = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
    tot = nchan * n
    for name, fn, b in (("matched filter", lambda: L.sdrx_psk_matched_filter_dev(vp(xq), n, nchan, n, vp(mf), m, st_()), 20.0),
                        ("timing + carrier loops", lambda: L.sdrx_psk_loops_dev(kind, vp(mf), m, nchan, m, vp(bank.states), vp(dec), bank.n_dec, 1, 1, st_()), 4.125),
                        ("ambiguity pass", lambda: L.sdrx_psk_ambiguity_dev(kind, vp(dec), (m + 1) // 2, nchan, bank.n_dec, vp(bank.states), st_()), 0.125),
                        ("whole chain (sdrx_psk_demod_dev)", lambda: bank.demod(xq, dec), 16.125)):
        t = timed(fn, iters=40, warm=10)
        rows.append({"kernel": f"cfg3 {tag} {name} (2^14 ch x 2^12)", "samples": tot, "ms": t * 1e3, "Msamples_per_s": tot / t / 1e6,
                     "algorithmic_GBps": b * tot / t / 1e9, "frac_of_8TBps": b * tot / t / 1e9 / PEAK})
    del xq, mf, dec, bank
print(json.dumps({"peak_GBps": PEAK, "rows": rows}, indent=1))
