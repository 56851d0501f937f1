import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from sdr_amd import _lib, device
device.init(0)
coef = device.notch_design(0.9, 0.05)
for nchan, n in ((131072, 1 << 10), (16384, 8192), (12288, 8192), (10240, 8192), (8192, 1 << 14), (4096, 1 << 15), (3072, 1 << 15), (2048, 1 << 16), (1024, 1 << 17), (256, 1 << 19), (64, 1 << 20)):
    x = torch.randn(nchan, n, dtype=torch.float64, device="cuda") * 50
    st = torch.zeros(nchan, 4, dtype=torch.float64, device="cuda")
    out = []
    for name, fn in (("dc", lambda: device.iir(device.IIR_DC_BLOCK, 0.995, x)), ("notch", lambda: device.iir(device.IIR_NOTCH, coef, x, state=st))):
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 5
        out.append(f"{name} {t:.3f} ms ({16.0 * nchan * n / t / 1e6 / 8000:.3f})")
    print(f"rows={nchan} n={n} default choice: " + ", ".join(out), flush=True)
    del x
