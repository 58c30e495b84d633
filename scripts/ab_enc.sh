cd if-defense_amd/csrc; cp libifd.so libifd_keep.so
for f in libifd_v_*.so; do cp $f libifd.so; printf "%-30s" $f; python ../../scripts/time_encoder.py 2>&1 | tail -1; done
mv libifd_keep.so libifd.so
