cd if-defense_amd/csrc; cp libifd.so libifd_keep.so
for f in libifd_v_*.so; do cp $f libifd.so; printf "%-24s" $f; python ../../scripts/time_optimize.py --clouds 256 --reps 3 2>&1 | grep "clock\|counters" | sed -e 's/.*-> \([0-9.]* k shader cycles per step\)/\1/' -e "s/.*'knn_rebuilds': \([0-9]*\).*'knn_ring_evals': \([0-9]*\), 'knn_exact_evals': \([0-9]*\).*/rebuilds \1 ring \2 exact \3/" | tr '\n' ' '; echo; done
mv libifd_keep.so libifd.so
