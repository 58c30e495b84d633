#!/bin/bash
# copies what scripts/collect_round5.sh left under gpurun_out/ into the tracked profiles/r05_* (run in the build container after the gpurun call)
cd "$(dirname "$0")/.."
cp gpurun_out/pmc_bench_r05.json profiles/r05_pmc_issue.json
for m in bf16x6 bf16x3; do
    cp gpurun_out/pmc_bench_r05_$m.json profiles/r05_pmc_issue_$m.json
    f=$(find gpurun_out/prof_r05_$m/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f profiles/r05_bench_kernel_stats_$m.csv
done
cp gpurun_out/r05_split_precision_traffic.json gpurun_out/r05_bench.json gpurun_out/r05_bench_streamed.json gpurun_out/r05_gpu_tests.log profiles/
cp gpurun_out/prof_r05/r05_bench_kernel_stats.csv gpurun_out/prof_r05/r05_pmc_fetch.csv gpurun_out/prof_r05/r05_pmc_tcc.csv gpurun_out/prof_r05/r05_pmc_write.csv profiles/
cp gpurun_out/prof_r05/roofline_traffic.json profiles/roofline_traffic.json
cp gpurun_out/prof_r05/roofline_traffic.json profiles/r05_roofline_traffic.json
cp gpurun_out/prof_r05/bench_under_rocprof.json profiles/r05_bench_under_rocprof.json
python - <<'P'
import json
b = json.load(open('profiles/r05_bench.json')); t = json.load(open('profiles/roofline_traffic.json'))
print("bench value", b["value"], "frac", b["roofline"]["frac"], "launch_ms", b["roofline"]["launch_ms"], "| rocprof avg ms", round(t["optimize_kernel_avg_ms"], 2), "sha", t["kernel_source_sha"])
ex = b["extras"]
print({k: (v.get("value") if isinstance(v, dict) else v) for k, v in ex.items()})
print("onet split", ex["onet_opt"].get("split_precision"), "mesh split", ex["onet_mesh"].get("split_precision"))
print("convonet split", {k: v["value"] for k, v in ex["split_precision"].items()})
P
