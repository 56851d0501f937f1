"""cfg3 stage timings (matched filter / loops / ambiguity / whole chain) under sdrx_tune settings.
usage: pskbench.py [name=value ...]"""
import ctypes as C, sys, os, json
sys.path.insert(0, os.getcwd())
import torch
from sdr_amd import _lib, device
L = _lib.lib()
device.init(0)
for a in sys.argv[1:]:
    if "=" in a and a.split("=")[0].startswith("SDRX_"):
        k, v = a.split("=")
        _lib.check(L.sdrx_tune(k.encode(), v.encode()), "tune")
opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a and not a.startswith("SDRX_"))
lc = int(opts.get("lc", 14)); kind = _lib.BPSK if opts.get("kind") == "bpsk" else _lib.QPSK
nchan, n = 1 << lc, 1 << (26 - lc)
g = torch.Generator(device="cuda").manual_seed(7)
x = torch.randn(nchan, n, 2, generator=g, device="cuda", dtype=torch.float64) * 0.3
bank = device.PskBank(kind, nchan, n)
out = torch.zeros(nchan, bank.n_dec, dtype=torch.uint8, device="cuda")
m = n // 4
mf = torch.zeros(nchan, m, 2, dtype=torch.float64, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def t(fn, w=10, k=40):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
res = {}
res["mf_ms"] = t(lambda: L.sdrx_psk_matched_filter_dev(p(x), n, nchan, n, p(mf), m, st()))
res["mf_kernel"] = L.sdrx_last_kernel().decode()
mf2 = mf.clone()
res["loops_ms"] = t(lambda: L.sdrx_psk_loops_dev(kind, p(mf2), m, nchan, m, p(bank.states), p(out), bank.n_dec, 1, 1, st()))
res["loops_kernel"] = L.sdrx_last_kernel().decode()
res["amb_ms"] = t(lambda: L.sdrx_psk_ambiguity_dev(kind, p(out), (m + 1) // 2, nchan, bank.n_dec, p(bank.states), st()))
res["chain_ms"] = t(lambda: bank.demod(x, out))
res["chain_kernel"] = L.sdrx_last_kernel().decode()
res["mf_ms_warm"] = t(lambda: L.sdrx_psk_matched_filter_dev(p(x), n, nchan, n, p(mf), m, st()), w=50, k=200)
res["Gsps"] = nchan * n / res["chain_ms"] / 1e6
print(json.dumps({"args": sys.argv[1:], **res}), flush=True)
