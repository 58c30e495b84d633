#!/bin/bash
# One-step wave trace with the -DIFD_TRACE build kept as if-defense_amd/csrc/libifd_trace.so (scripts/build_variant.sh trace "-DIFD_TRACE")
cd "$(dirname "$0")/../if-defense_amd/csrc"; cp libifd.so libifd_keep.so; cp libifd_trace.so libifd.so; cd ../..
python scripts/trace_step.py 2>&1 | grep -v amdgpu
cd if-defense_amd/csrc; mv libifd_keep.so libifd.so
