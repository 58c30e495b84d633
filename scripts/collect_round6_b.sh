#!/bin/bash
# Round 6, second GPU call: the launch-boundary question of bf16x6, the new large-cloud path (tests + per-point cost), Morton-ordered
# init points through the pipeline (headline command, short).
mkdir -p gpurun_out
timeout 600 python scripts/check_bf_launch_boundary.py bf16x6 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_bf_launch_boundary.txt
cat gpurun_out/r06_bf_launch_boundary.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "large or prepare or sweep or full_size or partial_round or streamed or cli" > gpurun_out/r06_gpu_tests_b.log 2>&1
tail -15 gpurun_out/r06_gpu_tests_b.log
grep "large lists\|large K=2048 t" gpurun_out/r06_gpu_tests_b.log | head -40
for scan in 1 0; do IFD_LARGE_SCAN=$scan timeout 300 python scripts/time_large_k.py 256 1024 2048 4096 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06_time_large_k.txt
IFD_LARGE_PRECISION=bf16x6 timeout 300 python scripts/time_large_k.py 256 1024 2048 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_time_large_k.txt
cat gpurun_out/r06_time_large_k.txt
timeout 600 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r06_bench_morton.json 2> gpurun_out/r06_bench_morton.err
cut -c1-600 gpurun_out/r06_bench_morton.json
