#!/bin/bash
# Per-phase cycle accounting with the -DIFD_PROF build kept as if-defense_amd/csrc/libifd_prof.so
#   IFD_EXTRA_FLAGS="-DIFD_PROF" python if-defense_amd/build.py --force && cp .../libifd.so .../libifd_prof.so && python if-defense_amd/build.py --force
cd "$(dirname "$0")/../if-defense_amd/csrc"; cp libifd.so libifd_keep.so; cp libifd_prof.so libifd.so; cd ../..
python scripts/time_pipeline_parts.py ${1:-256} 2>&1 | grep -v amdgpu
cd if-defense_amd/csrc; mv libifd_keep.so libifd.so
