import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from sdr_amd import _lib, device
device.init(0)
dev = torch.device("cuda", 0)
L = _lib.lib()
coef = device.notch_design(0.9, 0.05)
for nchan, n in ((16384, 1 << 13), (8192, 1 << 14), (131072, 1 << 10), (1024, 1 << 17), (65536, 1 << 11)):
    x = torch.randn(nchan, n, dtype=torch.float64, device=dev) * 50
    st = torch.zeros(nchan, 4, dtype=torch.float64, device=dev)
    for pair in ("0", "1", None):
        _lib.check(L.sdrx_tune(b"SDRX_IIR_PAIR", None if pair is None else pair.encode()), "tune")
        for name, fn in (("dc", lambda: device.iir(device.IIR_DC_BLOCK, 0.995, x)), ("notch", lambda: device.iir(device.IIR_NOTCH, coef, x, state=st))):
            for _ in range(3): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 10
            print(f"rows={nchan} n={n} pair={pair} {name}: {t:.3f} ms, {16.0 * nchan * n / t / 1e6 / 8000:.3f} of HBM", flush=True)
    del x
