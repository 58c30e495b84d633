#!/bin/bash
# A/B of optimiser-kernel builds (libifd_v_*.so) on the TRAINED-LIKE field: 512 bench clouds, kernel launch time + list counters
cd if-defense_amd/csrc; cp libifd.so libifd_keep.so
for f in libifd_v_*.so; do cp $f libifd.so; printf "%-24s " $f; IFD_SPLIT=1 IFD_WEIGHTS=trained python ../../scripts/time_pipeline_parts.py 512 2>&1 | grep "rep_weight   500" | tail -1 | cut -c1-150; done
mv libifd_keep.so libifd.so
