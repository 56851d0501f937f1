"""Raw-byte config 1 for other filter lengths: fir_amv_i8_bytes_kernel (runs of tiles per workgroup) against fir_amv_i8_kernel
(one tile per workgroup), 2^27 samples, one channel.  usage: python profiles/probes/amv_bytes_taps.py"""
import json, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import _oracle as O
from sdr_amd import _lib, device
device.init(0)
L = _lib.lib()
n = 1 << 27
g = torch.Generator(device="cuda").manual_seed(3)
raw = torch.randint(60, 196, (1, n, 2), generator=g, device="cuda", dtype=torch.uint8)
out = torch.zeros(1, n, dtype=torch.uint8, device="cuda")
res = {}
for T in (17, 48, 64, 65, 129, 200, 241):
    err, taps = O.fir_design(O.LPF, T, 2.0e6, 1.0e5)
    for by in (1, 0):
        L.sdrx_tune(b"SDRX_AMV_BYTES", str(by).encode())
        ch = device.FirChain(taps, nchan=1, decim=1)
        for _ in range(5): ch.fir_am(raw, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ch.fir_am(raw, out)
        e1.record(); torch.cuda.synchronize()
        res["T=%d %s" % (T, "bytes kernel" if by else "one tile per workgroup")] = {"ms": e0.elapsed_time(e1) / 20, "kernel": L.sdrx_last_kernel().decode()[:40], "ones": int(out.sum().item())}
print(json.dumps(res, indent=1))
