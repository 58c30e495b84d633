#!/bin/bash
# Issue-side PMC counters of the Winograd U-Net kernels (scripts/time_encoder.py, 512 clouds), summed over their dispatches.
#   bash scripts/pmc_wino.sh <tag>  -> gpurun_out/pmc_wino_<tag>.json      (separate rocprofv3 --pmc passes)
TAG=${1:-x}
R=$(pwd)
OUT=$R/gpurun_out/pmc_wino_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAVES"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python $R/scripts/time_encoder.py 512 > /dev/null 2>&1
done
cd $R
python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "wino_kernel" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    last = set(ids[-13:])                       # the 13 Winograd layers of the last launch_unet
    for r in rows:
        if int(r["Dispatch_Id"]) in last:
            k = r["Kernel_Name"].split("(")[0][-17:]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            acc["all"][r["Counter_Name"]] += float(r["Counter_Value"])
out = {}
for k, a in acc.items():
    a = dict(a)
    if a.get("SQ_INSTS_MFMA"):
        a["valu_non_mfma_per_mfma"] = (a.get("SQ_INSTS_VALU", 0) - a["SQ_INSTS_MFMA"]) / a["SQ_INSTS_MFMA"]
        a["lds_insts_per_mfma"] = a.get("SQ_INSTS_LDS", 0) / a["SQ_INSTS_MFMA"]
    if a.get("SQ_BUSY_CYCLES") and a.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        a["mfma_busy_of_busy_cycles"] = a["SQ_VALU_MFMA_BUSY_CYCLES"] / a["SQ_BUSY_CYCLES"]
    out[k] = a
json.dump(out, open("$R/gpurun_out/pmc_wino_$TAG.json", "w"), indent=1)
print(json.dumps(out["all"], indent=1))
PY
