#!/bin/bash
# rocprofv3 evidence for the cfg5 study (1024-tap bank): VALU exact / VALU fma chain / fp64 MFMA Toeplitz.
# Kernel trace and counters in separate runs; raw output under gpurun_out/prof/<tag>_cfg5_<variant>.
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
for v in exact fma mfma; do
  case $v in exact) FL="--exact";; fma) FL="--fast-fma";; mfma) FL="--mfma";; esac
  OUT=gpurun_out/prof/${TAG}_cfg5_$v
  mkdir -p $OUT
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o cfg5 --output-format csv -- python3 bench.py --workload cfg5 --steps 20 --warmup 5 --no-cpu $FL > $OUT/kt.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS -d $OUT/sq -o cfg5 --output-format csv -- python3 bench.py --workload cfg5 --steps 3 --warmup 2 --no-cpu $FL > $OUT/sq.log 2>&1 || exit 2
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o cfg5 --output-format csv -- python3 bench.py --workload cfg5 --steps 3 --warmup 2 --no-cpu $FL > $OUT/fetch.log 2>&1 || exit 3
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o cfg5 --output-format csv -- python3 bench.py --workload cfg5 --steps 3 --warmup 2 --no-cpu $FL > $OUT/write.log 2>&1 || exit 4
  grep -h '"metric"' $OUT/kt.log | tail -1 > $OUT/bench_line.json
done
echo collected
