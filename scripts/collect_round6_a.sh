#!/bin/bash
# Round 6, first GPU call: the whole GPU suite (parity matrix in both precisions), the point-order locality A/B, and the f32 kernel's
# step / tile traces + per-phase accounting (variants prebuilt in the build container: scripts/build_variant.sh trace "-DIFD_TRACE",
# trace2_1 / trace2_3 "-DIFD_TRACE -DIFD_TRACE2=<n>", prof "-DIFD_PROF").  Run from the repo root on the GPU box.
mkdir -p gpurun_out
R=$(pwd)
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r06_gpu_tests_first.log 2>&1
tail -5 gpurun_out/r06_gpu_tests_first.log
timeout 600 python scripts/ab_locality.py 2468 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_ab_bf_locality.txt
cat gpurun_out/r06_ab_bf_locality.txt
IFD_LIB=$R/if-defense_amd/csrc/libifd_v_trace.so timeout 300 python scripts/trace_step.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_trace.txt
for v in 1 3; do IFD_LIB=$R/if-defense_amd/csrc/libifd_v_trace2_$v.so timeout 300 python scripts/tile_trace.py 256 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_tile_trace_$v.txt; done
IFD_LIB=$R/if-defense_amd/csrc/libifd_v_prof.so timeout 300 python scripts/time_pipeline_parts.py 2468 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_prof.txt
cat gpurun_out/r06_prof.txt
