#!/bin/bash
# WRITE_SIZE / FETCH_SIZE counters and the kernel trace of am_modulate_kernel (is 7.45 TB/s real?)
export TMPDIR=/tmp
OUT=gpurun_out/prof/am_mod
mkdir -p $OUT
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $OUT/kt -o am --output-format csv -- python3 profiles/probes/am_mod_probe.py > $OUT/kt.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o am --output-format csv -- python3 profiles/probes/am_mod_probe.py > $OUT/write.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o am --output-format csv -- python3 profiles/probes/am_mod_probe.py > $OUT/fetch.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $OUT/wrreq -o am --output-format csv -- python3 profiles/probes/am_mod_probe.py > $OUT/wrreq.log 2>&1 || echo "wrreq pass failed (counter names?)"
python3 - <<PY
import csv, glob, json, collections
res = {}
for tag in ("write", "fetch", "wrreq"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if "am_modulate" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
for f in glob.glob("$OUT/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "am_modulate" in r["Name"]:
            res["kernel_trace"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
res["probe_line"] = [l.strip() for l in open("$OUT/kt.log") if l.startswith("am_modulate")]
json.dump(res, open("gpurun_out/am_mod_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
