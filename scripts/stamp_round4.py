#!/usr/bin/env python
"""Copy what `bash scripts/collect_round4.sh` left under gpurun_out/ into the tracked profiles/.   python scripts/stamp_round4.py"""
import os
import shutil

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
COPY = {"r04_bench.json": "r04_bench.json", "r04_bench_serial.json": "r04_bench_serial.json",
        "prof_r04/r04_bench_kernel_stats.csv": "r04_bench_kernel_stats.csv",
        "prof_r04/bench_under_rocprof.json": "r04_bench_under_rocprof.json",
        "prof_r04/r04_pmc_fetch.csv": "r04_pmc_fetch.csv", "prof_r04/r04_pmc_write.csv": "r04_pmc_write.csv",
        "prof_r04/r04_pmc_tcc.csv": "r04_pmc_tcc.csv", "prof_r04/roofline_traffic.json": "roofline_traffic.json",
        "pmc_bench_r04.json": "r04_pmc_issue.json", "r04_check_split.txt": "r04_check_split.txt", "r04_trace.txt": "r04_trace.txt",
        "r04_tile_trace_1.txt": "r04_tile_trace.txt", "r04_tile_trace_3.txt": "r04_tile_trace_last_tile.txt",
        "r04_prof.txt": "r04_prof.txt", "r04_unet_layers.txt": "r04_unet_layers.txt",
        "r04_onet_mesh_kernel_stats.txt": "r04_onet_mesh_kernel_stats.txt",
        "r04_strong_proxy_309.json": "r04_strong_proxy_309.json", "r04_strong_proxy_617.json": "r04_strong_proxy_617.json",
        "r04_ab_extra_valu.txt": "r04_ab_extra_valu.txt", "r04_ab_wv.txt": "r04_ab_ds_read_b32.txt", "r04_ab_relu.txt": "r04_ab_packed_relu.txt",
        "r04_ab_relu_equal.txt": "r04_ab_packed_relu_equal.txt", "r04_t3.log": "r04_attribution_tests.log", "r04_t5.log": "r04_trained_like_p1.log"}
for src, dst in COPY.items():
    if os.path.exists(os.path.join(G, src)):
        shutil.copyfile(os.path.join(G, src), os.path.join(P, dst))
    else:
        print("missing", src)
