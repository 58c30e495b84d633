#!/bin/bash
# Issue-side PMC counters of ifd::optimize_kernel ON THE BENCH WORKLOAD (the 2468-cloud launch of `bench.py --steps 1`:
# real encoder planes of the synthetic clouds), one JSON file.  Separate rocprofv3 --pmc passes, kernel trace only.
#   bash scripts/pmc_bench.sh <tag> [clouds]  -> gpurun_out/pmc_bench_<tag>.json
TAG=${1:-x}
CLOUDS=${2:-2468}
R=$(pwd)
OUT=$R/gpurun_out/pmc_bench_$TAG
CMD="python $R/bench.py --steps 1 --warmup 0 --clouds $CLOUDS --no-extras --no-cpu-baseline ${PMC_BENCH_ARGS:-}"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pass$i -o p -- $CMD > $OUT/pass$i.json 2> $OUT/pass$i.err
done
cd $R
python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(float)
clouds = $CLOUDS
for f in glob.glob("$OUT/pass*/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "ifd::optimize_kernel" in r["Kernel_Name"]]
    if not rows:
        continue
    big = max(int(r["Grid_Size"]) for r in rows)
    wg = next(int(r.get("Workgroup_Size", 512) or 512) for r in rows if int(r["Grid_Size"]) == big)
    clouds = big // wg                       # the largest launch of the file (with the partial round first: its whole rounds), one workgroup of 512 threads per cloud
    disp = sorted({r["Dispatch_Id"] for r in rows if int(r["Grid_Size"]) == big})[-1]
    for r in rows:
        if r["Dispatch_Id"] == disp:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
a = dict(acc)
a["clouds_in_the_measured_launch"] = clouds
waves = clouds * 8
if a.get("SQ_WAVE_CYCLES"):
    # SQ_WAVE_CYCLES counts in units of 4 cycles; two waves share a SIMD, so SIMD-resident cycles = wave cycles / 2
    a["cycles_per_step_per_wave"] = a["SQ_WAVE_CYCLES"] * 4 / waves / 501
    a["mfma_busy_frac"] = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["SQ_WAVE_CYCLES"] * 4 / 2) if a.get("SQ_VALU_MFMA_BUSY_CYCLES") else None
    a["valu_non_mfma_per_simd_step"] = (a.get("SQ_INSTS_VALU", 0) - a.get("SQ_INSTS_MFMA", 0)) / (clouds * 4) / 501
    a["valu_to_mfma"] = (a.get("SQ_INSTS_VALU", 0) - a.get("SQ_INSTS_MFMA", 0)) / max(1.0, a.get("SQ_INSTS_MFMA", 0))
    # pipe occupancy (round 4): f32 MFMA and the other vector instructions share one datapath on gfx950, so what the SIMDs can
    # be asked for is MFMA-busy cycles + issue cycles of the non-MFMA vector instructions (a wave64 instruction occupies the
    # 16-lane pipe for 4 cycles; packed / transcendental / DPP ones longer, so this is a lower bound) over SIMD cycles
    simd = a["SQ_WAVE_CYCLES"] * 4 / 2
    nv = a.get("SQ_INSTS_VALU", 0) - a.get("SQ_INSTS_MFMA", 0)
    a["valu_issue_frac_at_4_cycles"] = 4.0 * nv / simd
    a["pipe_occupancy_lower_bound"] = (a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) + 4.0 * nv) / simd
    a["lds_issue_frac"] = a.get("SQ_ACTIVE_INST_LDS", 0) * 4 / simd if a.get("SQ_ACTIVE_INST_LDS") else None
a["command"] = "bash scripts/pmc_bench.sh $TAG $CLOUDS: separate rocprofv3 --pmc passes over bench.py --steps 1 --warmup 0 --clouds $CLOUDS --no-extras --no-cpu-baseline --no-overlap (the largest optimize_kernel launch of the $CLOUDS-cloud file x 501 steps on the bench workload)"
json.dump(a, open("$R/gpurun_out/pmc_bench_$TAG.json", "w"), indent=1)
print(json.dumps(a, indent=1))
PY
