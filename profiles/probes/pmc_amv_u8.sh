#!/bin/bash
# PMC passes over bench.py cfg1 --ingest u8 (the int8 form on the radio's raw bytes): instruction mix and waits
export TMPDIR=/tmp
OUT=gpurun_out/prof/amv_u8
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT/a -o p --output-format csv -- python3 bench.py --workload cfg1 --ingest u8 --steps 5 --warmup 20 --no-cpu --no-extra > $OUT/log_a.txt 2>&1 || exit 4
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d $OUT/b -o p --output-format csv -- python3 bench.py --workload cfg1 --ingest u8 --steps 5 --warmup 20 --no-cpu --no-extra > $OUT/log_b.txt 2>&1 || exit 5
python3 - <<PY
import csv, collections
for sub in "ab":
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open("$OUT/%s/p_counter_collection.csv" % sub)):
        if "fir_amv_i8_bytes" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(k, sum(v) / len(v))
PY
