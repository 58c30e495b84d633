#!/bin/bash
# A/B of optimiser-kernel builds on ONE box (boxes differ by ~5 % in sustained clock, so only same-box numbers and
# the shader-cycle counts compare): every if-defense_amd/csrc/libifd_v_*.so is copied over libifd.so in turn and timed
# with scripts/time_optimize.py; the original libifd.so is restored at the end.
#   build variants:  IFD_EXTRA_FLAGS="-DX" python if-defense_amd/build.py --force && cp .../libifd.so .../libifd_v_X.so
cd "$(dirname "$0")/../if-defense_amd/csrc" || exit 1
cp libifd.so libifd_keep.so
for rep in 1 2; do
  for f in libifd_v_*.so; do
    cp "$f" libifd.so
    printf "%-48s " "$f"
    python ../../scripts/time_optimize.py --clouds ${CLOUDS:-256} --reps 3 2>&1 | grep "clouds ${CLOUDS:-256}\|clock" | tail -2 | \
      sed -e 's/.*: \([0-9.]* ms\).*| \([0-9.]* us\/step\).*/\1 \2/' -e 's/.*-> \([0-9.]* k shader cycles per step\)/\1/' | tr '\n' ' '
    echo
  done
done
mv libifd_keep.so libifd.so
