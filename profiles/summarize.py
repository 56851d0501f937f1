#!/usr/bin/env python3
"""Summarise gpurun_out/prof/<tag> (profiles/collect.sh) into profiles/<tag>_<workload>_*.

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half of the bytes of a wide
(16 B/lane) coalesced streaming read (MI355X_MICROARCH.md, section HBM): it is doubled here.
WRITE_SIZE is taken as is (exact for 16 B/lane stores; our 8 B/lane stores are uncalibrated but
small).  Values are per launch of the dominant kernel.
"""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
wl = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
kname = sys.argv[3] if len(sys.argv) > 3 else "fir_"
root = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(root, "..", "gpurun_out", "prof", tag)


def counters(sub):
    acc = collections.defaultdict(list)
    path = os.path.join(src, f"{sub}_{wl}", f"{wl}_counter_collection.csv")
    if not os.path.exists(path):  # round-1 layout: one directory per pass, shared by the workloads
        path = os.path.join(src, sub, f"{wl}_counter_collection.csv")
    if not os.path.exists(path):
        return {}
    for r in csv.DictReader(open(path)):
        if kname in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


stats = []
kt = os.path.join(src, f"kt_{wl}", f"{wl}_kernel_stats.csv")
if not os.path.exists(kt):
    kt = os.path.join(src, "kt", f"{wl}_kernel_stats.csv")
for r in csv.DictReader(open(kt)):
    stats.append(r)
with open(os.path.join(root, f"{tag}_{wl}_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in stats[:12]:
        w.writerow([r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
dom = next(r for r in stats if kname in r["Name"])
pmc = {}
for sub in ("fetch", "write", "sq", "sq2"):
    pmc.update(counters(sub))
fetch_b = 2.0 * pmc.get("FETCH_SIZE", 0.0) * 1024.0
write_b = pmc.get("WRITE_SIZE", 0.0) * 1024.0
out = {
    "tag": tag, "workload": wl, "kernel": dom["Name"][:120], "calls": int(dom["Calls"]),
    "avg_ns": float(dom["AverageNs"]), "min_ns": float(dom["MinNs"]), "max_ns": float(dom["MaxNs"]),
    "pmc_per_launch": pmc,
    "hbm_read_bytes_per_launch": fetch_b, "hbm_write_bytes_per_launch": write_b,
    "hbm_bytes_per_launch": fetch_b + write_b,
    "corrections": "FETCH_SIZE x2 (gfx950 wide-read half-count), KiB -> bytes; WRITE_SIZE KiB -> bytes",
}
bl = os.path.join(src, f"bench_line_{wl}.json")  # one per workload (round 1 shared one file: r01_cfg1 carried cfg3's line)
if os.path.exists(bl):
    try:
        out["bench_line_under_profiler"] = json.loads(open(bl).read())
    except Exception:
        pass
json.dump(out, open(os.path.join(root, f"{tag}_{wl}_summary.json"), "w"), indent=1)
json.dump({"hbm_bytes_per_launch": fetch_b + write_b, "source": f"profiles/{tag}_{wl}_summary.json"},
          open(os.path.join(root, f"pmc_traffic_{wl}.json"), "w"))
print(json.dumps({k: out[k] for k in ("kernel", "avg_ns", "hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch")}, indent=1))
for k, v in sorted(pmc.items()):
    print(f"  {k}: {v:.4g}")
