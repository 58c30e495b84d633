#!/bin/bash
# Build an optimiser-kernel variant for same-box A/B runs (scripts/ab_libs.sh):
#   scripts/build_variant.sh NAME "-DFLAG1 -DFLAG2"   ->  if-defense_amd/csrc/libifd_v_NAME.so
# Only optimize.hip is recompiled; the other objects of the last regular build (if-defense_amd/build.py) are linked as they are.
set -e
cd "$(dirname "$0")/../if-defense_amd/csrc"
NAME=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -I. -I../../include $* -x hip -c optimize.hip -o /tmp/optimize_$NAME.o 2>/dev/null
OBJS=$(ls *.o | grep -v '^optimize.o$' | grep -v '^optimize_exact.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libifd_v_$NAME.so /tmp/optimize_$NAME.o $OBJS
echo built libifd_v_$NAME.so
