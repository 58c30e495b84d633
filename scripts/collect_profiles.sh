#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline object refers to.  Run on the GPU box from the repo root:
#   bash scripts/collect_profiles.sh <tag>      -> gpurun_out/prof_<tag>/{stats,fetch,write,tcc}/...csv
# (kernel trace + stats in one run; every PMC set in its own run, never combined with sys/runtime tracing)
set -u
TAG=${1:-r01}
PREC=${2:-f32}                     # f32 | bf16x6 | bf16x3: which optimiser the passes run (bench.py --profile-precision)
EXTRA=""; [ "$PREC" != "f32" ] && EXTRA="--profile-precision $PREC"
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- \
    python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras $EXTRA > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
for set in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum"; do
    name=${set%%:*}; ctr=${set#*:}
    timeout 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/$name -o $name -- \
        python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras $EXTRA > /dev/null 2> $OUT/$name.err
done
# calibration of FETCH_SIZE on the decoder tile's gather pattern (a known byte count, every 128-byte line once)
if [ -x $R/scripts/gather_calib ]; then
    timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/calib -o calib -- \
        $R/scripts/gather_calib > $OUT/calib.json 2> $OUT/calib.err
fi
cd $R
python scripts/summarise_profiles.py $OUT $TAG $PREC
