import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from sdr_amd import _lib, device
device.init(0)
L = _lib.lib()
coef = device.notch_design(0.9, 0.05)
def tune(k, v): _lib.check(L.sdrx_tune(k, None if v is None else str(v).encode()), "tune")
for nchan, n in ((12288, 8192), (8192, 1 << 14), (4096, 1 << 15), (2048, 1 << 16), (1024, 1 << 17), (256, 1 << 19)):
    x = torch.randn(nchan, n, dtype=torch.float64, device="cuda") * 50
    st = torch.zeros(nchan, 4, dtype=torch.float64, device="cuda")
    for cw in (4, 16):
        for pair in (0, 1):
            tune(b"SDRX_IIR_CW", cw); tune(b"SDRX_IIR_PAIR", pair)
            for name, fn in (("dc", lambda: device.iir(device.IIR_DC_BLOCK, 0.995, x)), ("notch", lambda: device.iir(device.IIR_NOTCH, coef, x, state=st))):
                for _ in range(2): fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): fn()
                e1.record(); torch.cuda.synchronize()
                print(f"rows={nchan} n={n} cw={cw} pair={pair} {name}: {e0.elapsed_time(e1) / 5:.3f} ms", flush=True)
    del x
