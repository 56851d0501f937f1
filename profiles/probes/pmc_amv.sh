#!/bin/bash
# PMC pass over bench.py cfg1 for one form of the verified AM chain: $1 = SDRX_FIR_AMV (2 vector, 3 matrix), $2 = SDRX_AMV_NB
export TMPDIR=/tmp SDRX_FIR_AMV=$1 SDRX_AMV_NB=${2:-4}
OUT=gpurun_out/prof/amv_$1_$2
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT -o p --output-format csv -- python3 bench.py --workload cfg1 --steps 5 --warmup 20 --no-cpu --no-extra > $OUT/log.txt 2>&1 || exit 4
python3 - <<PY
import csv, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/p_counter_collection.csv")):
    if "fir_amv" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, sum(v) / len(v))
PY
grep -h '"metric"' $OUT/log.txt | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['roofline']['kernel'], d['roofline']['launch_ms'])"
