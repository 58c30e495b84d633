#!/bin/bash
# Which phase owns the LDS bank conflicts of ifd::optimize_kernel: the same launch (scripts/time_optimize.py, 256 clouds x 501
# steps) with the repulsion term on (kNN phase runs) and off (rep_weight 0: the kNN phase is skipped), one --pmc pass each.
#   bash scripts/pmc_lds_attrib.sh  -> gpurun_out/pmc_lds_attrib.txt
R=$(pwd)
OUT=$R/gpurun_out/pmc_lds_attrib
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in 500 0; do
    timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/w$w -o p -- \
        python $R/scripts/time_optimize.py --clouds 256 --reps 1 --rep_weight $w > $OUT/w$w.log 2>&1
done
cd $R
python - <<PY > gpurun_out/pmc_lds_attrib.txt
import csv, glob, collections
for w in (500, 0):
    acc = collections.defaultdict(float)
    n = 0
    for f in glob.glob("$OUT/w%d/**/*counter_collection.csv" % w, recursive=True):
        for r in csv.DictReader(open(f)):
            if "ifd::optimize_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                n += 1
    launches = max(1, n // 4)
    print("rep_weight", w, "launches", launches, {k: "%.1f per cloud-step" % (v / launches / 256 / 501) for k, v in acc.items()})
PY
cat gpurun_out/pmc_lds_attrib.txt
