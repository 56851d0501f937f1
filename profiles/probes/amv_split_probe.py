import json, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import _oracle as O
import bench
from sdr_amd import _lib, device
device.init(0)
L = _lib.lib()
n = 1 << 27
x = bench.synth_ook_stream(torch, n, 1, torch.device("cuda", 0))
re, im = x[:, :, 0].contiguous(), x[:, :, 1].contiguous()
out = torch.zeros(1, n, dtype=torch.uint8, device="cuda")
taps = bench.lpf_taps(64, 2.0e6, 1.0e5)
for name, inp in (("interleaved", x), ("split", (re, im))):
    for form in (1, 3, 0):
        L.sdrx_tune(b"SDRX_FIR_AMV", str(form).encode())
        ch = device.FirChain(taps, nchan=1, decim=1)
        for _ in range(5): ch.fir_am(inp, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ch.fir_am(inp, out)
        e1.record(); torch.cuda.synchronize()
        print(name, form, round(e0.elapsed_time(e1) / 20, 4), L.sdrx_last_kernel().decode()[:48], int(out.sum().item()))
