#!/usr/bin/env python3
"""cfg3's three kernels, round 2: run-time window walk (SDRX_FIR_CT=0, round 1's matched filter) against the
compile-time walk (default), from gpurun_out/prof/r02 (profiles/collect_r02.sh) -> profiles/r02_cfg3_chain.json.
Counters are averages per launch of the named kernel; times are rocprofv3 --kernel-trace --stats averages."""
import collections
import csv
import json
import os

root = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(root, "..", "gpurun_out", "prof", "r02")
KERNELS = {"matched_filter": "fir_tile_kernel", "loops": "psk_loops", "ambiguity": "psk_ambiguity"}


def stats(sub):
    out = {}
    for r in csv.DictReader(open(os.path.join(src, sub, "cfg3_kernel_stats.csv"))):
        for k, pat in KERNELS.items():
            if pat in r["Name"]:
                out[k] = {"name": r["Name"][:110], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                          "min_us": float(r["MinNs"]) / 1e3}
    return out


def counters(subs):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in subs:
        for r in csv.DictReader(open(os.path.join(src, sub, "cfg3_counter_collection.csv"))):
            for k, pat in KERNELS.items():
                if pat in r["Kernel_Name"]:
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


useful = (1 << 24) * 129 * 2 * 2 / 64  # outputs x taps x rails x (mul, add) / 64 lanes = wave-level VALU instructions
out = {"workload": "cfg3: 2^14 QPSK channels x 2^12 samples", "useful_valu_wave_insts_matched_filter": useful}
for tag, kt, pm in (("run_time_walk_SDRX_FIR_CT_0", "kt_cfg3_ct0", ("sq_cfg3_ct0", "sq2_cfg3_ct0")),
                    ("compile_time_walk_default", "kt_cfg3", ("sq_cfg3", "sq2_cfg3"))):
    s, c = stats(kt), counters(pm)
    for k in s:
        s[k]["pmc_per_launch"] = c.get(k, {})
    mf = s["matched_filter"]["pmc_per_launch"]
    s["matched_filter"]["valu_overhead"] = mf["SQ_INSTS_VALU"] / useful
    s["matched_filter"]["salu_per_valu"] = mf["SQ_INSTS_SALU"] / mf["SQ_INSTS_VALU"]
    s["chain_us"] = sum(v["avg_us"] for v in s.values() if isinstance(v, dict) and "avg_us" in v)
    out[tag] = s
json.dump(out, open(os.path.join(root, "r02_cfg3_chain.json"), "w"), indent=1)
for tag in ("run_time_walk_SDRX_FIR_CT_0", "compile_time_walk_default"):
    s = out[tag]
    mf = s["matched_filter"]
    print(tag, "chain %.1f us" % s["chain_us"], {k: round(v["avg_us"], 1) for k, v in s.items() if isinstance(v, dict)},
          "VALU %.3g SALU %.3g SMEM %.3g overhead %.2f" % (mf["pmc_per_launch"]["SQ_INSTS_VALU"], mf["pmc_per_launch"]["SQ_INSTS_SALU"],
                                                   mf["pmc_per_launch"].get("SQ_INSTS_SMEM", 0), mf["valu_overhead"]))
