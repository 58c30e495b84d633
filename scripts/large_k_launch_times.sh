#!/bin/bash
# Per-launch durations of the two kernels of the K > 1024 path (K = 2048, 501 steps) for 256 / 32 / 1 clouds: how much of the
# list-step launch is its slowest cloud (lock-step launches) and how much fixed cost.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 256 32 1; do
  rm -rf /tmp/lk_$n
  IFD_LARGE_STEPS=501 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lk_$n -o lk -- python $R/scripts/time_large_k.py $n 2048 > /tmp/lk_$n.log 2>&1
  echo "== $n clouds"; grep "^K =" /tmp/lk_$n.log || tail -5 /tmp/lk_$n.log
  f=$(find /tmp/lk_$n -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 1.0:
        print("   %-60s calls %6s  avg %9.1f us  min %9.1f  max %9.1f  share %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
PY
done
