#!/bin/bash
# Stream groups of the K > 1024 path (api.cpp large_optimize_in_groups): K = 2048, 501 steps, time per point over the persistent
# kernel's at the same cloud count; ONE group (measurement hook) against the automatic choice.  [points ...] = digest of the result.
export IFD_LARGE_STEPS=501
run() { echo "== $1 clouds, ${2:-automatic} group(s)"; IFD_ENABLE_TEST_HOOKS=1 IFD_TEST_LARGE_GROUPS=${2:-0} python scripts/time_large_k.py $1 1024 2048 2>&1 | grep "^K =  2048" | sed 's/; 1.0 list epochs.*//'; }
for n in 64 256 512 1024 2304; do run $n 1; run $n; done
echo "== K = 4096, 256 clouds"
IFD_ENABLE_TEST_HOOKS=1 IFD_TEST_LARGE_GROUPS=1 python scripts/time_large_k.py 256 1024 4096 2>&1 | grep "^K =  4096" | sed 's/; 1.0 list epochs.*//'
python scripts/time_large_k.py 256 1024 4096 2>&1 | grep "^K =  4096" | sed 's/; 1.0 list epochs.*//'
echo "== bf16x6, 256 clouds"
IFD_LARGE_PRECISION=bf16x6 python scripts/time_large_k.py 256 1024 2048 2>&1 | grep "^K"
