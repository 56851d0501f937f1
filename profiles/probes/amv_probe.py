import sys, time, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch, ctypes as C
from sdr_amd import _lib, device
import bench
device.init(0)
L = _lib.lib()
dev = torch.device("cuda", 0)
taps = bench.lpf_taps(64, 2.0e6, 1.0e5)

def run(x, amv, stats=False):
    L.sdrx_tune(b"SDRX_FIR_AMV", b"1" if amv else b"0")
    L.sdrx_tune(b"SDRX_AMV_STATS", b"1" if stats else b"0")
    ch = device.FirChain(taps, nchan=x.shape[0], decim=1, device=dev)
    out = torch.zeros(x.shape[0], x.shape[1], dtype=torch.uint8, device=dev)
    if stats:
        L.sdrx_fir_am_verify_stats(None, 1)
    ch.fir_am(x, out)
    torch.cuda.synchronize()
    st = None
    if stats:
        a = (C.c_uint64 * 4)()
        L.sdrx_fir_am_verify_stats(a, 1)
        st = list(a)
    return out, ch.state[ch.cur].clone(), L.sdrx_last_kernel().decode(), st

def ook(n, seed, amp=100.0, spb=8, noise=10.0):
    g = torch.Generator(device=dev).manual_seed(seed)
    bits = torch.randint(0, 2, ((n + spb - 1) // spb,), generator=g, device=dev).to(torch.float64)
    env = bits.repeat_interleave(spb)[:n] * amp
    ph = torch.rand(1, generator=g, device=dev, dtype=torch.float64) * 6.28
    x = torch.empty(1, n, 2, dtype=torch.float64, device=dev)
    x[0, :, 0] = torch.round(env * torch.cos(ph) + (torch.rand(n, generator=g, device=dev, dtype=torch.float64) * 2 - 1) * noise)
    x[0, :, 1] = torch.round(env * torch.sin(ph) + (torch.rand(n, generator=g, device=dev, dtype=torch.float64) * 2 - 1) * noise)
    return x

res = {}
for name, x in [("ook_2^22", ook(1 << 22, 1)), ("ook_ragged", ook((1 << 20) + 12345, 2)), ("tiny", ook(37, 3)),
                ("fm", bench.synth_fm_stream(torch, 1 << 22, 5, dev)),
                ("randn*60", torch.randn(3, 100000, 2, device=dev, dtype=torch.float64) * 60)]:
    o1, s1, k1, st = run(x, True, True)
    o0, s0, k0, _ = run(x, False)
    res[name] = {"equal": bool(torch.equal(o0, o1)), "state_equal": bool(torch.equal(s0, s1)), "ones": int(o1.sum()),
                 "n": x.shape[0] * x.shape[1], "stats": st, "k1": k1, "k0": k0}
    print(name, res[name], flush=True)

# timing on cfg1's 2^28
for gen in ("ook", "fm"):
    n = 1 << 28
    x = ook(n, 7) if gen == "ook" else bench.synth_fm_stream(torch, n, 1000, dev)
    for amv in (1, 0):
        L.sdrx_tune(b"SDRX_FIR_AMV", b"1" if amv else b"0")
        L.sdrx_tune(b"SDRX_AMV_STATS", b"0")
        ch = device.FirChain(taps, nchan=1, decim=1, device=dev)
        out = torch.zeros(1, n, dtype=torch.uint8, device=dev)
        for _ in range(10): ch.fir_am(x, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): ch.fir_am(x, out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        print(gen, "amv" if amv else "exact", "ms/launch", round(ms, 4), "Gs/s", round(n / ms / 1e6, 1), "frac_hbm", round(17 * n / ms / 1e6 / 8000, 3), L.sdrx_last_kernel().decode(), flush=True)
    if gen == "ook":
        L.sdrx_tune(b"SDRX_FIR_AMV", b"1"); L.sdrx_tune(b"SDRX_AMV_STATS", b"1")
        L.sdrx_fir_am_verify_stats(None, 1)
        ch = device.FirChain(taps, nchan=1, decim=1, device=dev)
        ch.fir_am(x, out); a = (C.c_uint64 * 4)(); L.sdrx_fir_am_verify_stats(a, 1)
        print("stats 2^28 ook", list(a), "waves", n // 512, flush=True)
    del x, out
