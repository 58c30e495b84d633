#!/bin/bash
# copies what scripts/collect_round6.sh left under gpurun_out/ into the tracked profiles/r06_* (run in the build container after the gpurun call)
cd "$(dirname "$0")/.."
python scripts/summarise_profiles.py gpurun_out/prof_r06 r06 f32 > /dev/null
python scripts/summarise_profiles.py gpurun_out/prof_r06_bf16x6 r06_bf16x6 bf16x6 > /dev/null
cp gpurun_out/prof_r06/roofline_traffic.json profiles/roofline_traffic.json
cp gpurun_out/prof_r06/roofline_traffic.json profiles/r06_roofline_traffic.json
cp gpurun_out/prof_r06_bf16x6/roofline_traffic.json profiles/roofline_traffic_bf16x6.json
cp gpurun_out/prof_r06_bf16x6/roofline_traffic.json profiles/r06_roofline_traffic_bf16x6.json
cp gpurun_out/prof_r06/r06_bench_kernel_stats.csv gpurun_out/prof_r06/r06_pmc_fetch.csv gpurun_out/prof_r06/r06_pmc_tcc.csv gpurun_out/prof_r06/r06_pmc_write.csv profiles/
cp gpurun_out/prof_r06_bf16x6/r06_bf16x6_bench_kernel_stats.csv gpurun_out/prof_r06_bf16x6/r06_bf16x6_pmc_fetch.csv gpurun_out/prof_r06_bf16x6/r06_bf16x6_pmc_tcc.csv gpurun_out/prof_r06_bf16x6/r06_bf16x6_pmc_write.csv profiles/
cp gpurun_out/prof_r06/bench_under_rocprof.json profiles/r06_bench_under_rocprof.json
cp gpurun_out/prof_r06_bf16x6/bench_under_rocprof.json profiles/r06_bench_under_rocprof_bf16x6.json
cp gpurun_out/pmc_bench_r06.json profiles/r06_pmc_issue.json
cp gpurun_out/pmc_bench_r06_bf16x6.json profiles/r06_pmc_issue_bf16x6.json
cp gpurun_out/r06_ab_morton_pipeline.txt gpurun_out/r06_strong_scaling_proxies.jsonl gpurun_out/r06_bench_streamed.json profiles/
for f in r06_bench.json r06_gpu_tests.log; do [ -f gpurun_out/$f ] && cp gpurun_out/$f profiles/; done
python - <<'P'
import json
t = json.load(open('profiles/roofline_traffic.json')); t6 = json.load(open('profiles/roofline_traffic_bf16x6.json'))
print("f32: rocprof avg ms", round(t["optimize_kernel_avg_ms"], 2), "per cloud GB", round(t["per_cloud_bytes"] / 1e9, 3), "L2 hit", round(t["L2_hit_rate"], 3), "sha", t["kernel_source_sha"])
print("bf16x6: rocprof avg ms", round(t6["optimize_kernel_avg_ms"], 2), "per cloud GB", round(t6["per_cloud_bytes"] / 1e9, 3), "L2 hit", round(t6["L2_hit_rate"], 3), "sha", t6["kernel_source_sha"])
P
