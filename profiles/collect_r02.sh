#!/bin/bash
# Round-2 evidence, one gpurun call:  bash profiles/collect_r02.sh
#   * collect.sh for every BASELINE config (kernel trace of the driver's own steps/warmup + separate PMC passes)
#   * the matched filter of cfg3 with the run-time window walk (SDRX_FIR_CT=0, round 1's kernel) for the before/after
#     SQ_INSTS_VALU / SQ_INSTS_SALU comparison VERDICT r01 item 4(b) asks for
set -u
export TMPDIR=/tmp
for wl in cfg2 cfg3 cfg1 cfg4 cfg5; do
    bash profiles/collect.sh r02 $wl || exit $?
done
OUT=gpurun_out/prof/r02
export SDRX_FIR_CT=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_cfg3_ct0 -o cfg3 --output-format csv -- python3 bench.py --workload cfg3 --steps 20 --warmup 5 --no-cpu --no-extra > $OUT/kt_cfg3_ct0.log 2>&1 || exit 11
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU -d $OUT/sq_cfg3_ct0 -o cfg3 --output-format csv -- python3 bench.py --workload cfg3 --steps 5 --warmup 20 --no-cpu --no-extra > $OUT/sq_cfg3_ct0.log 2>&1 || exit 12
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d $OUT/sq2_cfg3_ct0 -o cfg3 --output-format csv -- python3 bench.py --workload cfg3 --steps 5 --warmup 20 --no-cpu --no-extra > $OUT/sq2_cfg3_ct0.log 2>&1 || exit 13
echo collected r02
