"""Fallback counters of the verified AM chain on bench.py's cfg1 input (2^28 samples): outputs recomputed in fp64, waves that
recomputed any, tiles whose bound was unusable, tiles that took the int8 matrix path.  One launch per form."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench
from sdr_amd import _lib, device
device.init(0)
L = _lib.lib()
dev = torch.device("cuda", 0)
taps = bench.lpf_taps(64, 2.0e6, 1.0e5)
n = 1 << 28
x = bench.synth_ook_stream(torch, n, 1, dev)
out = torch.zeros(1, n, dtype=torch.uint8, device=dev)
res = {"samples": n, "input": "bench.py synth_ook_stream (cfg1)"}
L.sdrx_tune(b"SDRX_AMV_STATS", b"1")
for form, name in ((1, "default"), (4, "int8"), (3, "fp32_mfma"), (2, "fp32_fma")):
    L.sdrx_tune(b"SDRX_FIR_AMV", str(form).encode())
    ch = device.FirChain(taps, nchan=1, decim=1, device=dev)
    L.sdrx_fir_am_verify_stats(None, 1)
    ch.fir_am(x, out)
    a = (C.c_uint64 * 4)()
    L.sdrx_fir_am_verify_stats(a, 1)
    res[name] = {"kernel": L.sdrx_last_kernel().decode(), "outputs_recomputed": int(a[0]), "waves_that_recomputed": int(a[1]),
                 "tiles_with_unusable_bound": int(a[2]), "tiles_on_the_int8_path": int(a[3]), "ones": int(out.sum().item()),
                 "recomputed_fraction": int(a[0]) / n}
print(json.dumps(res, indent=1))
