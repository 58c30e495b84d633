#!/bin/bash
# Everything profiles/r05_* of the final kernels comes from, in one gpurun call (run from the repo root on the GPU box; ~20 min)
mkdir -p gpurun_out
R=$(pwd)
bash scripts/collect_profiles.sh r05 > gpurun_out/collect_r05.log 2>&1
bash scripts/pmc_bench.sh r05 > gpurun_out/pmc_bench_r05.log 2>&1
python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
python bench.py --streamed --no-extras --no-cpu-baseline > gpurun_out/r05_bench_streamed.json 2>/dev/null
# the opt-in split-precision modes: issue-side counters and the HBM traffic of their launches (the gather-bound claim of DESIGN 4.3)
for m in bf16x6 bf16x3; do
    PMC_BENCH_ARGS="--profile-precision $m" bash scripts/pmc_bench.sh r05_$m > gpurun_out/pmc_bench_r05_$m.log 2>&1
done
cd /tmp && export TMPDIR=/tmp
for m in bf16x6 bf16x3; do
    O=$R/gpurun_out/prof_r05_$m; mkdir -p $O
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o stats -- \
        python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --profile-precision $m > $O/bench_under_rocprof.json 2> $O/stats.err
    for set in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum"; do
        name=${set%%:*}; ctr=${set#*:}
        timeout 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/$name -o $name -- \
            python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --profile-precision $m > /dev/null 2> $O/$name.err
    done
done
cd $R
python - <<'P' > gpurun_out/r05_split_precision_traffic.json
import csv, glob, json
out = {}
for m in ("bf16x6", "bf16x3"):
    d = "gpurun_out/prof_r05_%s" % m
    acc = {}
    for sub in ("fetch", "write", "tcc"):
        for f in glob.glob("%s/%s/**/*counter_collection.csv" % (d, sub), recursive=True):
            for r in csv.DictReader(open(f)):
                if "optimize_kernel" in r["Kernel_Name"] and "onet" not in r["Kernel_Name"]:
                    acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    st = {}
    for f in glob.glob("%s/stats/**/*kernel_stats.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "optimize_kernel" in r["Name"] and "onet" not in r["Name"]:
                st = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6, "share_pct": float(r["Percentage"])}
    try:
        factor = json.load(open("gpurun_out/prof_r05/roofline_traffic.json")).get("fetch_factor", 2.0)
    except Exception:
        factor = 2.0
    e = {"counters": acc, "kernel_stats": st, "fetch_factor": factor}
    if acc.get("FETCH_SIZE") and st:
        # as scripts/summarise_profiles.py: read bytes = calibrated fetch factor x FETCH_SIZE x 1024, WRITE_SIZE at face value
        rd = acc["FETCH_SIZE"] * 1024.0 * factor
        wr = acc.get("WRITE_SIZE", 0.0) * 1024.0
        e["hbm_read_bytes_per_launch"] = rd
        e["hbm_write_bytes_per_launch"] = wr
        e["hbm_GBps_over_the_launch"] = (rd + wr) / (st["avg_ms"] * 1e-3) / 1e9
    out[m] = e
print(json.dumps(out, indent=1))
P
cut -c1-400 gpurun_out/r05_bench.json
