"""Does the order of measurements matter?  cfg2's driver-style measurement (5 warm-up + 20 timed launches) (a) first thing in
the process, (b) again right after, (c) after one second of another workload's launches, (d) after 0.5 s of host sleep."""
import sys, os, time, json
sys.path.insert(0, os.getcwd())
import torch
import bench
from sdr_amd import device
device.init(0)
dev = torch.device("cuda", 0)
a = bench.parse_args(["--workload", "cfg2"])
w = bench.build_workload("cfg2", a, 0, 1, dev)
res = {}
def measure(tag):
    dt, ev = bench.timed_steps(torch, w["step"], 5, 20)
    res[tag] = ev / 20
    print(tag, round(ev / 20, 4), "ms per launch", flush=True)
measure("a_first_thing")
measure("b_again")
a4 = bench.parse_args(["--workload", "cfg4", "--log2n", "18"])
w4 = bench.build_workload("cfg4", a4, 0, 1, dev)
t0 = time.time()
while time.time() - t0 < 1.0:
    for _ in range(20): w4["step"]()
    torch.cuda.synchronize()
measure("c_after_1s_of_cfg4")
time.sleep(0.5)
measure("d_after_half_a_second_idle")
dt, ev = bench.timed_steps(torch, w["step"], 50, 200)
res["e_50_warm_200_timed"] = ev / 200
print("e", ev / 200, flush=True)
print(json.dumps(res))
