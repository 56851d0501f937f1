"""am_modulate at the shape of profiles/kernel_rates.py (1024 channels x 2^14 bits x 16 samples per bit = 2 GiB of doubles
written per launch), a few launches: run under rocprofv3 --kernel-trace and under --pmc WRITE_SIZE (profiles/probes/am_mod_pmc.sh)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from sdr_amd import device
device.init(0)
dev = torch.device("cuda", 0)
nchan, nbit, spb = 1024, 1 << 14, 16
bits = torch.randint(0, 2, (nchan, nbit), device=dev, dtype=torch.uint8)
mod = torch.zeros(nchan, nbit * spb, dtype=torch.float64, device=dev)
for _ in range(3): device.am_modulate(bits, mod)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): device.am_modulate(bits, mod)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("am_modulate ms per launch", ms, "GB/s written", 8.0 * nchan * nbit * spb / ms / 1e6, "bytes per launch", 8 * nchan * nbit * spb, flush=True)
