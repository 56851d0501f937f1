"""Engine clock (rocm-smi) while each BASELINE config's kernel runs back to back."""
import sys, os, subprocess, time, re, json
sys.path.insert(0, os.getcwd())
import torch
import bench
from sdr_amd import _lib, device
device.init(0)
dev = torch.device("cuda", 0)
def smi():
    o = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True).stdout
    s = re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", o); m = re.findall(r"mclk clock level: \d+: \((\d+)Mhz\)", o)
    return int(s[0]) if s else None, int(m[0]) if m else None
res = {}
for name, flags in (("cfg2", []), ("cfg1", []), ("cfg5", []), ("cfg5_mfma", ["--mfma"]), ("cfg5_fma", ["--fast-fma"])):
    a = bench.parse_args(["--workload", name.split("_")[0]] + flags)
    w = bench.build_workload(a.workload, a, 0, 1, dev)
    for _ in range(5): w["step"]()
    torch.cuda.synchronize()
    samples = []
    t_end = time.time() + 2.0
    while time.time() < t_end:
        for _ in range(8 if name.startswith("cfg5") else 100): w["step"]()
        samples.append(smi())
    torch.cuda.synchronize()
    res[name] = samples[-3:]
    print(name, samples[-3:], flush=True)
    del w
    torch.cuda.empty_cache()
print(json.dumps(res))
