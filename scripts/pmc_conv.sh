#!/bin/bash
# Issue-side PMC counters of ifd::optimize_kernel only (scripts/time_optimize.py, 256 clouds x 501 steps), one JSON line.
#   bash scripts/pmc_conv.sh <tag>  -> gpurun_out/pmc_conv_<tag>.json      (separate rocprofv3 --pmc passes)
TAG=${1:-x}
R=$(pwd)
OUT=$R/gpurun_out/pmc_conv_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/conv$i -o p -- python $R/scripts/time_optimize.py --clouds 256 --reps 1 > /dev/null 2>&1
done
cd $R
python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(float)
for f in glob.glob("$OUT/conv*/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "ifd::optimize_kernel" in r["Kernel_Name"]]
    if not rows:
        continue
    big = max(int(r["Grid_Size"]) for r in rows)
    disp = sorted({r["Dispatch_Id"] for r in rows if int(r["Grid_Size"]) == big})[-1]
    for r in rows:
        if r["Dispatch_Id"] == disp:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
a = dict(acc)
if a.get("SQ_WAVE_CYCLES"):
    a["cycles_per_step_per_wave"] = a["SQ_WAVE_CYCLES"] * 4 / 2048 / 501
    a["mfma_busy_frac"] = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * a["SQ_WAVE_CYCLES"] * 4 / 2048) if a.get("SQ_VALU_MFMA_BUSY_CYCLES") else None
    a["valu_non_mfma_per_simd_step"] = (a.get("SQ_INSTS_VALU", 0) - a.get("SQ_INSTS_MFMA", 0)) / 1024 / 501
json.dump(a, open("$R/gpurun_out/pmc_conv_$TAG.json", "w"), indent=1)
print(json.dumps(a, indent=1))
PY
