import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from sdr_amd import _lib, device
device.init(0)
dev = torch.device("cuda", 0)
def timed(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
def tune(name, v):
    _lib.check(_lib.lib().sdrx_tune(name.encode(), None if v is None else str(v).encode()), "tune")
for n in (16, 64, 128, 256, 512, 1024, 2048, 4096):
    rows = (1 << 26) // n
    z = torch.randn(rows, n, dtype=torch.complex128, device=dev)
    for ct in (1, 0):
        tune("SDRX_FFT_CT", ct)
        t = timed(lambda: device.fft(z))
        print(f"fft {n} x {rows} ct {ct}: {t:.3f} ms, {32.0 * rows * n / t / 1e6 / 8000:.3f} of HBM", flush=True)
