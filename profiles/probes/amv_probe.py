import sys, time, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch, ctypes as C
from sdr_amd import _lib, device
import bench
device.init(0)
L = _lib.lib()
dev = torch.device("cuda", 0)
taps = bench.lpf_taps(64, 2.0e6, 1.0e5)

def run(x, amv, stats=False, taps=taps):
    L.sdrx_tune(b"SDRX_FIR_AMV", str(int(amv)).encode())
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
for name, x in [] if 'time' in sys.argv else [("ook_2^22", ook(1 << 22, 1)), ("ook_ragged", ook((1 << 20) + 12345, 2)), ("tiny", ook(37, 3)),
                ("fm", bench.synth_fm_stream(torch, 1 << 22, 5, dev)),
                ("randn*60", torch.randn(3, 100000, 2, device=dev, dtype=torch.float64) * 60)]:
    o0, s0, k0, _ = run(x, 0)
    for mode in (2, 3, 13):
        L.sdrx_tune(b'SDRX_AMV_NB', b'2' if mode == 13 else b'4')
        o1, s1, k1, st = run(x, mode % 10, True)
        res[name] = {"equal": bool(torch.equal(o0, o1)), "state_equal": bool(torch.equal(s0, s1)), "ones": int(o1.sum()),
                     "n": x.shape[0] * x.shape[1], "stats": st, "k1": k1[:26]}
        print(name, mode, res[name], flush=True)

for T in () if 'time' in sys.argv else (1, 2, 5, 7, 8, 9, 63, 65, 129, 1000):
    tp = bench.lpf_taps(T, 2.0e6, 1.0e5) if T > 2 else np.array([0.9, -0.4][:T])
    if T > 2: tp = tp * (1.0 / np.abs(tp).sum()) * 1.3
    x = ook((1 << 18) + 777, 100 + T)
    o0, s0, k0, _ = run(x, 0, taps=tp)
    for mode in (2, 3, 13):
        L.sdrx_tune(b'SDRX_AMV_NB', b'2' if mode == 13 else b'4')
        o1, s1, k1, st = run(x, mode % 10, True, taps=tp)
        print("T", T, mode, bool(torch.equal(o0, o1)), bool(torch.equal(s0, s1)), int(o1.sum()), st, k1[:22], flush=True)
# split rails
x = ook((1 << 20) + 3, 55)
outs = []
for amv in (0, 2, 3):
    L.sdrx_tune(b"SDRX_FIR_AMV", str(amv).encode())
    ch = device.FirChain(taps, nchan=1, decim=1, device=dev)
    out = torch.zeros(1, x.shape[1], dtype=torch.uint8, device=dev)
    ch.fir_am((x[:, :, 0].contiguous(), x[:, :, 1].contiguous()), out)
    torch.cuda.synchronize()
    print("split", amv, int(out.sum()), L.sdrx_last_kernel().decode(), flush=True)
    outs.append(out)
print("split equal", bool(torch.equal(outs[0], outs[1])), bool(torch.equal(outs[0], outs[2])), flush=True)

# timing on cfg1's 2^28
for gen in ("ook", "fm"):
    n = 1 << 28
    x = ook(n, 7) if gen == "ook" else bench.synth_fm_stream(torch, n, 1000, dev)
    for amv, nb in ((4, 2), (1, 2), (3, 4), (3, 2), (2, 4), (0, 4)):
        L.sdrx_tune(b"SDRX_FIR_AMV", str(amv).encode())
        L.sdrx_tune(b"SDRX_AMV_NB", str(nb).encode())
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
        print(gen, amv, "ms/launch", round(ms, 4), "Gs/s", round(n / ms / 1e6, 1), "frac_hbm", round(17 * n / ms / 1e6 / 8000, 3), L.sdrx_last_kernel().decode(), flush=True)
    if gen == "ook":
        L.sdrx_tune(b"SDRX_FIR_AMV", b"3"); L.sdrx_tune(b"SDRX_AMV_STATS", b"1")
        L.sdrx_fir_am_verify_stats(None, 1)
        ch = device.FirChain(taps, nchan=1, decim=1, device=dev)
        ch.fir_am(x, out); a = (C.c_uint64 * 4)(); L.sdrx_fir_am_verify_stats(a, 1)
        print("stats 2^28 ook", list(a), "waves", n // 512, flush=True)
    del x, out
