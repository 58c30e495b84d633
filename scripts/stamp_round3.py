#!/usr/bin/env python
"""Copy what `bash scripts/collect_round3.sh` left under gpurun_out/ into the tracked profiles/ (names as profiles/README.md
lists them) and assemble profiles/r03_strong_scaling_proxies.json from the eight proxy bench lines.   python scripts/stamp_round3.py"""
import json
import os
import shutil

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
COPY = {"r03_bench.json": "r03_bench.json", "r03_bench_serial.json": "r03_bench_serial.json",
        "prof_r03/r03_bench_kernel_stats.csv": "r03_bench_kernel_stats.csv",
        "prof_r03/bench_under_rocprof.json": "r03_bench_under_rocprof.json",
        "prof_r03/r03_pmc_fetch.csv": "r03_pmc_fetch.csv", "prof_r03/r03_pmc_write.csv": "r03_pmc_write.csv",
        "prof_r03/r03_pmc_tcc.csv": "r03_pmc_tcc.csv", "prof_r03/roofline_traffic.json": "roofline_traffic.json",
        "pmc_bench_r03.json": "r03_pmc_issue.json", "r03_ab_planes.txt": "r03_ab_planes.txt",
        "r03_check_split.txt": "r03_check_split.txt", "r03_trace.txt": "r03_trace.txt",
        "r03_prof_trained.txt": "r03_prof_trained.txt", "r03_prof.txt": "r03_prof.txt"}
for src, dst in COPY.items():
    if os.path.exists(os.path.join(G, src)):
        shutil.copyfile(os.path.join(G, src), os.path.join(P, dst))
    else:
        print("missing", src)


def line(name):
    return json.loads(open(os.path.join(G, name)).read().strip().splitlines()[-1])


rows = []
full = line("r03_strong_proxy_2468.json")["value"]
for gpus, n in ((1, 2468), (2, 1234), (4, 617), (8, 309)):
    a, b = line("r03_strong_proxy_%d.json" % n), line("r03_strong_proxy_nosplit_%d.json" % n)
    rows.append({"gpus_it_stands_for": gpus, "clouds": n, "clouds_per_s": a["value"], "ms_per_pass": a["ms_per_step"],
                 "optimiser_launch_ms": a["roofline"]["launch_ms"], "fraction_of_full_file_rate": round(a["value"] / full, 3),
                 "nosplit_clouds_per_s": b["value"], "nosplit_fraction": round(b["value"] / full, 3)})
json.dump({"what": "single-GPU proxies of the strong-scaling configurations (one 2468-cloud file over N GPUs -> 2468/N clouds per "
                   "GPU), same box, bench.py --clouds n --steps 5 --warmup 1 --no-extras --no-cpu-baseline; nosplit = IFD_SPLIT=1 "
                   "(one workgroup per cloud throughout, the round-2 behaviour)", "rows": rows},
          open(os.path.join(P, "r03_strong_scaling_proxies.json"), "w"), indent=1)
for r in rows:
    print(r)
