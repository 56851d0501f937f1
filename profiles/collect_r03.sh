#!/bin/bash
# Round-3 evidence, one gpurun call:  bash profiles/collect_r03.sh [workloads...]
#   * collect.sh for every BASELINE config (kernel trace of a driver-style run + separate PMC passes)
#   * cfg1: one more pass with the matrix-core counters of the verified AM chain, and its fallback counters
set -u
export TMPDIR=/tmp
WLS=${@:-cfg2 cfg3 cfg1 cfg4 cfg5}
for wl in $WLS; do
    bash profiles/collect.sh r03 $wl || exit $?
done
OUT=gpurun_out/prof/r03
case " $WLS " in *" cfg1 "*)
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT/mfma_cfg1 -o cfg1 --output-format csv -- python3 bench.py --workload cfg1 --steps 5 --warmup 20 --no-cpu --no-extra > $OUT/mfma_cfg1.log 2>&1 || exit 21
SDRX_AMV_STATS=1 timeout -k 10 120 python3 profiles/probes/amv_stats.py > $OUT/amv_stats_cfg1.json 2> $OUT/amv_stats_cfg1.err || exit 22
;; esac
echo collected r03
