#!/bin/bash
# Round 6, fifth GPU call: large-cloud path with quad-cooperative list renewal (tests, per-point cost, kernel by kernel)
mkdir -p gpurun_out
R=$(pwd)
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "large" > gpurun_out/r06_gpu_tests_e.log 2>&1
tail -4 gpurun_out/r06_gpu_tests_e.log
grep "large lists" gpurun_out/r06_gpu_tests_e.log | head -20
for scan in 1 0; do IFD_LARGE_SCAN=$scan timeout 300 python scripts/time_large_k.py 256 1024 2048 4096 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06_time_large_k4.txt
IFD_LARGE_STEPS=501 timeout 300 python scripts/time_large_k.py 256 1024 2048 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_time_large_k4.txt
cat gpurun_out/r06_time_large_k4.txt
cd /tmp && export TMPDIR=/tmp
for scan in 0; do
    rm -rf /tmp/lk$scan
    IFD_LARGE_SCAN=$scan timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lk$scan -o lk -- python $R/scripts/time_large_k.py 256 1024 2048 > /tmp/lk$scan.log 2>&1
    f=$(find /tmp/lk$scan -name "*kernel_stats.csv" | head -1)
    echo "== IFD_LARGE_SCAN=$scan"; python -c "
import csv,sys
for r in list(csv.DictReader(open('$f')))[:6]:
    print('%-60s calls %5s avg %10.1f us  min %10.1f max %10.1f  %5s %%' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, r['Percentage']))
"
done > $R/gpurun_out/r06_large_kernel_stats2.txt 2>&1
cat $R/gpurun_out/r06_large_kernel_stats2.txt
