#!/bin/bash
# A/B of optimiser-kernel builds on the BENCH workload (real encoder planes; the kNN dynamics differ from the synthetic planes
# of time_optimize.py): every libifd_v_*.so in turn, serial passes, kernel-only launch time from bench.py's HIP events.
cd "$(dirname "$0")/../if-defense_amd/csrc" || exit 1
cp libifd.so libifd_keep.so
for rep in 1 2; do
  for f in libifd_v_*.so; do
    cp "$f" libifd.so
    printf "%-32s " "$f"
    timeout 120 python ../../bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-overlap 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('launch_ms', j['roofline']['launch_ms'], 'frac', j['roofline']['frac'])"
  done
done
mv libifd_keep.so libifd.so
