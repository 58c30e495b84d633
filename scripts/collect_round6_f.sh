#!/bin/bash
mkdir -p gpurun_out
R=$(pwd)
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "large" > gpurun_out/r06_gpu_tests_f.log 2>&1
tail -3 gpurun_out/r06_gpu_tests_f.log
grep "large lists" gpurun_out/r06_gpu_tests_f.log | head -12
(timeout 300 python scripts/time_large_k.py 256 1024 2048 4096; IFD_LARGE_STEPS=501 timeout 300 python scripts/time_large_k.py 256 1024 2048 4096
 echo "== -DIFD_PROF build"; IFD_LARGE_STEPS=501 IFD_LIB=$R/if-defense_amd/csrc/libifd_v_prof.so timeout 300 python scripts/time_large_k.py 256 1024 2048 4096) 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_time_large_k10.txt
cat gpurun_out/r06_time_large_k10.txt
