#!/bin/bash
# Instruction-cache behaviour of ifd::optimize_kernel on the bench workload (the kernel is 190 KB of code, the I-cache 64 KB per two CUs):
#   bash scripts/pmc_icache.sh [clouds]  -> gpurun_out/r04_pmc_icache.txt
CLOUDS=${1:-512}
R=$(pwd)
OUT=$R/gpurun_out/pmc_icache
CMD="python $R/bench.py --steps 1 --warmup 0 --clouds $CLOUDS --no-extras --no-cpu-baseline --no-overlap"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQC_TC_INST_REQ SQC_ICACHE_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pass$i -o p -- $CMD > $OUT/pass$i.json 2> $OUT/pass$i.err
done
cd $R
python - <<PY | tee $R/gpurun_out/r04_pmc_icache.txt
import csv, glob, collections
acc = collections.defaultdict(float)
for f in glob.glob("$OUT/pass*/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "ifd::optimize_kernel" in r["Kernel_Name"]]
    if not rows: continue
    big = max(int(r["Grid_Size"]) for r in rows)
    disp = sorted({r["Dispatch_Id"] for r in rows if int(r["Grid_Size"]) == big})[-1]
    for r in rows:
        if r["Dispatch_Id"] == disp: acc[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(acc): print("%-32s %.4g" % (k, acc[k]))
n = $CLOUDS
if acc.get("SQC_ICACHE_REQ"):
    print("I-cache: %.2f %% of the requests miss (%.2f %% incl. duplicates); %.3g misses per cloud-step" % (
        100 * acc["SQC_ICACHE_MISSES"] / acc["SQC_ICACHE_REQ"], 100 * (acc["SQC_ICACHE_MISSES"] + acc["SQC_ICACHE_MISSES_DUPLICATE"]) / acc["SQC_ICACHE_REQ"],
        acc["SQC_ICACHE_MISSES"] / n / 501))
if acc.get("SQ_IFETCH"):
    print("instruction fetches per cloud-step %.4g, mean fetch latency (SQ_IFETCH_LEVEL / SQ_IFETCH) %.1f" % (acc["SQ_IFETCH"] / n / 501, acc["SQ_IFETCH_LEVEL"] / acc["SQ_IFETCH"]))
PY
