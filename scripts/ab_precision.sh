#!/bin/bash
# Same-box A/B of split-precision kernel variants: every if-defense_amd/csrc/libifd_v_*.so in turn through scripts/time_precision.py
# (256 clouds x 501 steps, synthetic planes), two passes.   scripts/ab_precision.sh [precision ...]
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for f in if-defense_amd/csrc/libifd_v_*.so; do
    printf "%-28s " "$(basename $f)"
    IFD_LIB=$PWD/$f python scripts/time_precision.py ${@:-bf16x6} 2>&1 | grep "planes own    rep_weight   500" | sed 's/planes own    rep_weight   500: //'
  done
done
