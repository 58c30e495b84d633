#!/bin/bash
# Build a variant of the split-precision optimiser kernels for same-box A/B runs:
#   scripts/build_variant_bf.sh NAME "-DFLAG1 -DFLAG2"   ->  if-defense_amd/csrc/libifd_v_NAME.so
# Only optimize_bf.hip is recompiled; the other objects of the last regular build are linked as they are.
set -e
cd "$(dirname "$0")/../if-defense_amd/csrc"
NAME=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -I. -I../../include -fno-slp-vectorize $* -x hip -c optimize_bf.hip -o /tmp/optimize_bf_$NAME.o 2>/dev/null
OBJS=$(ls *.o | grep -v '^optimize_bf.o$' | grep -v '^optimize_exact.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libifd_v_$NAME.so /tmp/optimize_bf_$NAME.o $OBJS
echo built libifd_v_$NAME.so
