"""Engine clock and power while one kernel runs back to back: matched filter, loops, cfg2 chain."""
import ctypes as C, sys, os, subprocess, time, json, re
sys.path.insert(0, os.getcwd())
import torch
from sdr_amd import _lib, device
L = _lib.lib(); device.init(0)
kind = _lib.QPSK
nchan, n = 1 << 14, 1 << 12
x = torch.randn(nchan, n, 2, device="cuda", dtype=torch.float64) * 0.3
bank = device.PskBank(kind, nchan, n)
out = torch.zeros(nchan, bank.n_dec, dtype=torch.uint8, device="cuda")
m = n // 4
mf = torch.zeros(nchan, m, 2, dtype=torch.float64, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def smi():
    o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    sclk = re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", o)
    pw = re.findall(r"Power \(W\): ([0-9.]+)", o)
    return (sclk[:1], pw[:1])
fns = {"matched_filter": lambda: L.sdrx_psk_matched_filter_dev(p(x), n, nchan, n, p(mf), m, st()),
       "loops": lambda: L.sdrx_psk_loops_dev(kind, p(mf), m, nchan, m, p(bank.states), p(out), bank.n_dec, 1, 1, st()),
       "chain": lambda: bank.demod(x, out)}
res = {}
for name, fn in fns.items():
    samples = []
    t_end = time.time() + 2.5
    k = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() < t_end:
        for _ in range(200): fn()
        k += 200
        samples.append(smi())
    e1.record(); torch.cuda.synchronize()
    res[name] = {"ms_per_call": e0.elapsed_time(e1) / k, "smi": samples[-4:]}
    print(name, res[name], flush=True)
print(json.dumps(res))
