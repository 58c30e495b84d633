#!/bin/bash
# Queue-side PMC counters (LDS / texture-addresser FIFOs full, MFMA + VALU co-execution) of the Winograd U-Net kernels and of the
# optimiser kernel, summed per kernel.   bash scripts/pmc_fifo.sh <tag>  -> gpurun_out/pmc_fifo_<tag>.txt   (separate --pmc passes)
TAG=${1:-x}
R=$(pwd)
OUT=$R/gpurun_out/pmc_fifo_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/e$i -o p -- python $R/scripts/time_encoder.py 512 > /dev/null 2>&1
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/o$i -o p -- python $R/bench.py --steps 1 --warmup 0 --clouds 256 --no-extras --no-cpu-baseline --no-overlap > /dev/null 2>&1
done
cd $R
python - <<PY > $R/gpurun_out/pmc_fifo_$TAG.txt
import csv, glob, collections
for pat, key, nlast in (("$OUT/e*", "wino_kernel", 14), ("$OUT/o*", "optimize_kernel", 1)):
    acc = collections.defaultdict(float)
    for f in glob.glob(pat + "/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if key in r["Kernel_Name"]]
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})
        last = set(ids[-nlast:])
        for r in rows:
            if int(r["Dispatch_Id"]) in last:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
    print(key)
    for k in sorted(acc):
        print("   %-32s %.4g" % (k, acc[k]))
    if acc.get("SQ_WAVE_CYCLES"):
        for k in acc:
            if k not in ("SQ_WAVE_CYCLES",):
                print("   %-32s / SQ_WAVE_CYCLES = %.4f" % (k, acc[k] / acc["SQ_WAVE_CYCLES"]))
PY
cat $R/gpurun_out/pmc_fifo_$TAG.txt
