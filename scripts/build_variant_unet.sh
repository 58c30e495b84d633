#!/bin/bash
# Build a U-Net variant for same-box A/B runs (scripts/ab_enc.sh):
#   scripts/build_variant_unet.sh NAME "-DFLAG1 -DFLAG2"   ->  if-defense_amd/csrc/libifd_v_NAME.so
# Only unet.hip is recompiled; the other objects of the last regular build (if-defense_amd/build.py) are linked as they are.
set -e
cd "$(dirname "$0")/../if-defense_amd/csrc"
NAME=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -I. -I../../include $* -x hip -c unet.hip -o /tmp/unet_$NAME.o 2>/dev/null
OBJS=$(ls *.o | grep -v '^unet.o$' | grep -v '^optimize_exact.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libifd_v_$NAME.so /tmp/unet_$NAME.o $OBJS
echo built libifd_v_$NAME.so
