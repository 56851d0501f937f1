import sys
sys.path.insert(0, "/root/repo")
import torch
from sdr_amd import _lib, device
import bench
device.init(0)
L = _lib.lib()
dev = torch.device("cuda", 0)
taps = bench.lpf_taps(64, 2.0e6, 1.0e5)
n = 1 << 28
x = bench.synth_ook_stream(torch, n, 1, dev)
out = torch.zeros(1, n, dtype=torch.uint8, device=dev)
import itertools
cases = [(3, 2, 0, d) for d in (0, 1, 2, 3, 4)] + [(3, 4, 0, 0), (2, 4, 0, 0), (0, 4, 0, 0)]
for mode, nb, pers, dbg in cases:
    if True:
        L.sdrx_tune(b"SDRX_FIR_AMV", str(mode).encode()); L.sdrx_tune(b"SDRX_AMV_NB", str(nb).encode()); L.sdrx_tune(b"SDRX_AMV_DBG", str(dbg).encode())
        ch = device.FirChain(taps, nchan=1, decim=1, device=dev)
        for _ in range(10): ch.fir_am(x, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): ch.fir_am(x, out)
        e1.record(); torch.cuda.synchronize()
        print("mode", mode, "NB", nb, "persist", pers, "dbg", dbg, "(skip:", "mfma" if dbg & 1 else "", "loads" if dbg & 2 else "", "decide" if dbg & 4 else "", ") ms", round(e0.elapsed_time(e1) / 30, 4), flush=True)
