set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/fftpmc; mkdir -p $OUT
timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE -d $OUT/fs -o f --output-format csv -- python3 $R/profiles/probes/fftone.py 2048 > $OUT/fs.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE -d $OUT/ws -o f --output-format csv -- python3 $R/profiles/probes/fftone.py 2048 > $OUT/ws.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU -d $OUT/sq1 -o f --output-format csv -- python3 $R/profiles/probes/fftone.py 2048 > $OUT/sq1.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE -d $OUT/sq2 -o f --output-format csv -- python3 $R/profiles/probes/fftone.py 2048 > $OUT/sq2.log 2>&1
python3 - <<'P'
import csv, glob, collections, os
for d in ("fs", "ws", "sq1", "sq2"):
    for f in glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/fftpmc", d, "**/*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "fft_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items(): print(d, k, sum(v) / len(v), len(v))
P
