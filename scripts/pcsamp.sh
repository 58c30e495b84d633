#!/bin/bash
# Per-instruction issue / stall histogram of ifd::optimize_kernel on the bench workload: rocprofv3 PC sampling (stochastic, then
# host-trap) and an Advanced Thread Trace attempt, each in its own run; whatever fails leaves its exact error text behind.
#   scripts/build_variant.sh lines "-gline-tables-only"      (here: source lines for every instruction, same code)
#   [IFD_TRY_ATT=1] bash scripts/pcsamp.sh <tag> [clouds] [interval]         (GPU box)  -> gpurun_out/<tag>_pcsamp_*.{json,txt,err}
TAG=${1:-r04}
CLOUDS=${2:-256}
IVAL=${3:-1048576}
R=$(pwd)
OUT=$R/gpurun_out/${TAG}_pcsamp
mkdir -p $OUT
cd $R/if-defense_amd/csrc && cp libifd.so libifd_keep.so && cp libifd_v_lines.so libifd.so && cd $R
CMD="python $R/bench.py --steps 1 --warmup 0 --clouds $CLOUDS --no-extras --no-cpu-baseline --no-overlap"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/avail.txt 2>&1
rocprofv3 --list-avail >> $OUT/avail.txt 2>&1
grep -i -A8 "pc.sampl" $OUT/avail.txt | head -60 > $R/gpurun_out/${TAG}_pcsamp_configs.txt

echo "== stochastic, cycles, interval $IVAL" > $R/gpurun_out/${TAG}_pcsamp_status.txt
timeout 900 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval $IVAL \
    --kernel-trace --output-format csv -d $OUT/stoch -o pc -- $CMD > $OUT/stoch.out 2> $OUT/stoch.err
echo "exit $?" >> $R/gpurun_out/${TAG}_pcsamp_status.txt
tail -c 3000 $OUT/stoch.err > $R/gpurun_out/${TAG}_pcsamp_stochastic.err
find $OUT/stoch -name "*.csv" -exec ls -la {} \; >> $R/gpurun_out/${TAG}_pcsamp_status.txt
for f in $(find $OUT/stoch -name "*pc_sampling*.csv"); do head -5 $f > $R/gpurun_out/${TAG}_pcsamp_stochastic_head.csv; done
python $R/scripts/pcsamp_hist.py $OUT/stoch $R/gpurun_out/${TAG}_pcsamp_stochastic.json $R/gpurun_out/${TAG}_pcsamp_stochastic.txt >> $R/gpurun_out/${TAG}_pcsamp_status.txt 2>&1

echo "== host_trap, time, interval 1 (us)" >> $R/gpurun_out/${TAG}_pcsamp_status.txt
timeout 900 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 1 \
    --kernel-trace --output-format csv -d $OUT/trap -o pc -- $CMD > $OUT/trap.out 2> $OUT/trap.err
echo "exit $?" >> $R/gpurun_out/${TAG}_pcsamp_status.txt
tail -c 3000 $OUT/trap.err > $R/gpurun_out/${TAG}_pcsamp_host_trap.err
find $OUT/trap -name "*.csv" -exec ls -la {} \; >> $R/gpurun_out/${TAG}_pcsamp_status.txt
for f in $(find $OUT/trap -name "*pc_sampling*.csv"); do head -5 $f > $R/gpurun_out/${TAG}_pcsamp_host_trap_head.csv; done
python $R/scripts/pcsamp_hist.py $OUT/trap $R/gpurun_out/${TAG}_pcsamp_host_trap.json $R/gpurun_out/${TAG}_pcsamp_host_trap.txt >> $R/gpurun_out/${TAG}_pcsamp_status.txt 2>&1

if [ -n "$IFD_TRY_ATT" ]; then
echo "== ATT (one CU, optimize_kernel only)" >> $R/gpurun_out/${TAG}_pcsamp_status.txt
timeout 600 rocprofv3 --att --att-target-cu 1 --att-buffer-size 0x40000000 --kernel-include-regex optimize_kernel --kernel-trace \
    -d $OUT/att -o att -- python $R/bench.py --steps 1 --warmup 0 --clouds 64 --no-extras --no-cpu-baseline --no-overlap > $OUT/att.out 2> $OUT/att.err
echo "exit $?" >> $R/gpurun_out/${TAG}_pcsamp_status.txt
tail -c 3000 $OUT/att.err > $R/gpurun_out/${TAG}_att.err
find $OUT/att -type f | head -40 >> $R/gpurun_out/${TAG}_pcsamp_status.txt
du -sh $OUT/att >> $R/gpurun_out/${TAG}_pcsamp_status.txt 2>&1
fi

cd $R/if-defense_amd/csrc && mv libifd_keep.so libifd.so
# the raw sample files are large: keep the histograms, drop the rest
rm -rf $OUT
cat $R/gpurun_out/${TAG}_pcsamp_status.txt
