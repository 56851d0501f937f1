#!/bin/bash
# Collect the rocprofv3 evidence for bench.py's default workload (run on the GPU box via gpurun):
#   1. --kernel-trace --stats of the exact default bench command
#   2. separate --pmc passes for FETCH_SIZE, WRITE_SIZE and SQ counters (never combined with traces)
# Raw output goes to gpurun_out/prof/<tag>; profiles/summarize.py turns it into the committed summaries.
set -u
TAG=${1:-r02}
WL=${2:-cfg2}
export TMPDIR=/tmp
OUT=gpurun_out/prof/$TAG
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_$WL -o $WL --output-format csv -- python3 bench.py --workload $WL --steps 20 --warmup 5 --no-cpu --no-extra > $OUT/kt_$WL.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch_$WL -o $WL --output-format csv -- python3 bench.py --workload $WL --steps 5 --warmup 20 --no-cpu --no-extra > $OUT/fetch_$WL.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $OUT/write_$WL -o $WL --output-format csv -- python3 bench.py --workload $WL --steps 5 --warmup 20 --no-cpu --no-extra > $OUT/write_$WL.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU -d $OUT/sq_$WL -o $WL --output-format csv -- python3 bench.py --workload $WL --steps 5 --warmup 20 --no-cpu --no-extra > $OUT/sq_$WL.log 2>&1 || exit 4
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d $OUT/sq2_$WL -o $WL --output-format csv -- python3 bench.py --workload $WL --steps 5 --warmup 20 --no-cpu --no-extra > $OUT/sq2_$WL.log 2>&1 || exit 5
grep -h '"metric"' $OUT/kt_$WL.log | tail -1 > $OUT/bench_line_$WL.json
echo collected $OUT
