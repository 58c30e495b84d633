#!/bin/bash
# Issue-side PMC evidence (MFMA busy, VALU instructions, wait cycles) for the two optimiser kernels.
#   bash scripts/collect_issue_pmc.sh <tag>   -> gpurun_out/pmc_issue_<tag>.json   (separate --pmc passes)
set -u
TAG=${1:-r01}
R=$(pwd)
OUT=$R/gpurun_out/pmc_issue_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/conv$i -o p -- python $R/scripts/time_optimize.py --clouds 256 --reps 1 > /dev/null 2>&1
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/onet$i -o p -- python $R/scripts/time_onet.py 256 11 > /dev/null 2>&1
done
cd $R
python - <<PY
import csv, glob, json, collections
res = {}
for kind, pat in (("ifd::optimize_kernel", "conv"), ("ifd::onet_optimize_kernel", "onet")):
    acc = collections.defaultdict(float)
    for f in glob.glob("$OUT/%s*/**/*counter_collection.csv" % pat, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if kind in r["Kernel_Name"]]
        if not rows:
            continue
        big = max(int(r["Grid_Size"]) for r in rows)
        disp = sorted({r["Dispatch_Id"] for r in rows if int(r["Grid_Size"]) == big})[-1]      # the last full-size launch
        for r in rows:
            if r["Dispatch_Id"] == disp:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
    res[kind.split("::")[1]] = dict(acc)
json.dump(res, open("$R/gpurun_out/pmc_issue_$TAG.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
