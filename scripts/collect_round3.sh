mkdir -p gpurun_out
bash scripts/collect_profiles.sh r03 > gpurun_out/collect_r03.log 2>&1
bash scripts/pmc_bench.sh r03 > gpurun_out/pmc_bench_r03.log 2>&1
python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
python bench.py --no-overlap --no-extras --no-cpu-baseline > gpurun_out/r03_bench_serial.json 2>/dev/null
for n in 309 617 1234 2468; do python bench.py --clouds $n --no-extras --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/r03_strong_proxy_$n.json 2>/dev/null; IFD_SPLIT=1 python bench.py --clouds $n --no-extras --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/r03_strong_proxy_nosplit_$n.json 2>/dev/null; done
python scripts/ab_planes.py 2468 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_ab_planes.txt
bash scripts/run_trace.sh > gpurun_out/r03_trace.txt 2>&1
bash scripts/run_prof.sh 2468 > gpurun_out/r03_prof.txt 2>&1
IFD_WEIGHTS=trained bash scripts/run_prof.sh 512 > gpurun_out/r03_prof_trained.txt 2>&1
python scripts/check_split.py 53 501 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_check_split.txt
cut -c1-400 gpurun_out/r03_bench.json
bash scripts/pmc_lds_attrib.sh > /dev/null 2>&1
