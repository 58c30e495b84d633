#!/bin/bash
# Round 6, third GPU call: the large-cloud path again (lists without epochs), kernel by kernel; the two bf16x6 test fixes; counters on offer.
mkdir -p gpurun_out
R=$(pwd)
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "large or config5 or trained_like_decoder or normal" > gpurun_out/r06_gpu_tests_c.log 2>&1
tail -8 gpurun_out/r06_gpu_tests_c.log
grep "large lists" gpurun_out/r06_gpu_tests_c.log | head -20
for scan in 1 0; do IFD_LARGE_SCAN=$scan timeout 300 python scripts/time_large_k.py 256 1024 2048 4096 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06_time_large_k2.txt
cat gpurun_out/r06_time_large_k2.txt
cd /tmp && export TMPDIR=/tmp
for scan in 1 0; do
    rm -rf /tmp/lk$scan
    IFD_LARGE_SCAN=$scan timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lk$scan -o lk -- python $R/scripts/time_large_k.py 256 1024 2048 > /tmp/lk$scan.log 2>&1
    f=$(find /tmp/lk$scan -name "*kernel_stats.csv" | head -1)
    echo "== IFD_LARGE_SCAN=$scan"; head -8 $f | cut -c1-60,200-400
done > $R/gpurun_out/r06_large_kernel_stats.txt 2>&1
cat $R/gpurun_out/r06_large_kernel_stats.txt
rocprofv3 --list-avail 2>/dev/null | grep -i -E "mall|dram|hbm|EA0?_RD|FETCH|WRITE_SIZE" | head -40 > $R/gpurun_out/r06_counters_on_offer.txt
cat $R/gpurun_out/r06_counters_on_offer.txt
