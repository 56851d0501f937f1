import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from sdr_amd import _lib, device
device.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
z = torch.randn((1 << 26) // n, n, dtype=torch.complex128, device="cuda")
for _ in range(8): device.fft(z)
torch.cuda.synchronize()
