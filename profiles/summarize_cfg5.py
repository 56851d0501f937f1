#!/usr/bin/env python3
"""Summarise gpurun_out/prof/<tag>_cfg5_{exact,fma,mfma} (profiles/collect_cfg5.sh) into
profiles/<tag>_cfg5_study.json: per variant the dominant kernel's average duration, TFLOP/s
(2*T flop per real sample), the PMC counters per launch and MFMA / VALU busy fractions."""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.abspath(__file__))
T, units = 1024, 1024 * (1 << 18)
out = {"tag": tag, "workload": "cfg5: 1024 channels x 2^18 real samples, 1024 taps, fp64", "flop_per_launch": 2.0 * T * units,
       "peak_tflops": 78.6, "variants": {}}
for v in ("exact", "fma", "mfma"):
    src = os.path.join(root, "..", "gpurun_out", "prof", f"{tag}_cfg5_{v}")
    st = list(csv.DictReader(open(os.path.join(src, "kt", "cfg5_kernel_stats.csv"))))
    dom = next(r for r in st if "fir_" in r["Name"])
    acc = collections.defaultdict(list)
    for sub in ("sq", "fetch", "write"):
        p = os.path.join(src, sub, "cfg5_counter_collection.csv")
        if os.path.exists(p):
            for r in csv.DictReader(open(p)):
                if "fir_" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    pmc = {k: sum(x) / len(x) for k, x in acc.items()}
    avg_ns = float(dom["AverageNs"])
    d = {"kernel": dom["Name"][:100], "calls": int(dom["Calls"]), "avg_ns": avg_ns,
         "tflops": out["flop_per_launch"] / avg_ns / 1e3, "pmc_per_launch": pmc,
         "hbm_read_bytes": 2.0 * pmc.get("FETCH_SIZE", 0) * 1024, "hbm_write_bytes": pmc.get("WRITE_SIZE", 0) * 1024}
    d["frac_of_peak"] = d["tflops"] / out["peak_tflops"]
    cu = pmc.get("SQ_BUSY_CU_CYCLES", 0)
    if cu:
        d["mfma_busy_frac"] = pmc.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * cu)  # 4 SIMDs per CU
        d["valu_issue_frac"] = 4 * pmc.get("SQ_ACTIVE_INST_VALU", 0) / (4 * cu)   # quad-cycles -> cycles, per SIMD
    bl = os.path.join(src, "bench_line.json")
    if os.path.exists(bl):
        try:
            d["bench_line_under_profiler"] = json.loads(open(bl).read())
        except Exception:
            pass
    out["variants"][v] = d
json.dump(out, open(os.path.join(root, f"{tag}_cfg5_study.json"), "w"), indent=1)
for v, d in out["variants"].items():
    print(v, "%.2f ms  %.1f TFLOP/s  frac %.3f  mfma_busy %.2f  valu_issue %.2f" % (
        d["avg_ns"] / 1e6, d["tflops"], d["frac_of_peak"], d.get("mfma_busy_frac", 0), d.get("valu_issue_frac", 0)))
