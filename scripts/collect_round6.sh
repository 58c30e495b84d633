#!/bin/bash
# Everything profiles/r06_* of the final kernels comes from, in one gpurun call (run from the repo root on the GPU box; ~25 min).
# scripts/stamp_round6.sh copies the results into profiles/ afterwards (build container).
mkdir -p gpurun_out
R=$(pwd)
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r06_gpu_tests.log 2>&1
tail -3 gpurun_out/r06_gpu_tests.log
# rocprofv3: kernel statistics + the HBM-side traffic of the optimiser, per launch shape, f32 and bf16x6 (separate --pmc passes)
bash scripts/collect_profiles.sh r06 > gpurun_out/collect_r06.log 2>&1
bash scripts/collect_profiles.sh r06_bf16x6 bf16x6 > gpurun_out/collect_r06_bf16x6.log 2>&1
# issue-side counters of the file's largest launch
bash scripts/pmc_bench.sh r06 > gpurun_out/pmc_bench_r06.log 2>&1
PMC_BENCH_ARGS="--profile-precision bf16x6" bash scripts/pmc_bench.sh r06_bf16x6 > gpurun_out/pmc_bench_r06_bf16x6.log 2>&1
# Morton-ordered initial points through the WHOLE pipeline, same box: the headline command with and without (measurement hook)
(for rep in 1 2; do for prec in f32 bf16x6; do for nm in 0 1; do
    echo "== precision $prec, IFD_TEST_NO_MORTON=$nm (rep $rep)"
    IFD_ENABLE_TEST_HOOKS=1 IFD_TEST_NO_MORTON=$nm python bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline --profile-precision $prec 2>/dev/null | \
        python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.1f clouds/s, ms per file %.1f, optimiser launches %s' % (d['value'], d['ms_per_step'], [(s['clouds'], s['ms']) for s in d['roofline']['launch_shapes']]))"
done; done; done) > gpurun_out/r06_ab_morton_pipeline.txt 2>&1
cat gpurun_out/r06_ab_morton_pipeline.txt
# single-GPU PROXIES of one GPU's share of a file spread over 8 / 4 GPUs (309 / 617 clouds): not a multi-GPU measurement
(for prec in f32 bf16x6; do for n in 309 617 2468; do
    python bench.py --clouds $n --steps 6 --warmup 2 --no-extras --no-cpu-baseline --profile-precision $prec 2>/dev/null | \
        python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'precision': '$prec', 'clouds': $n, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'launch_shapes': d['roofline']['launch_shapes']}))"
done; done) > gpurun_out/r06_strong_scaling_proxies.jsonl 2>&1
cat gpurun_out/r06_strong_scaling_proxies.jsonl
python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
python bench.py --streamed --no-extras --no-cpu-baseline > gpurun_out/r06_bench_streamed.json 2>/dev/null
IFD_LIB=$R/if-defense_amd/csrc/libifd_v_prof.so timeout 300 python scripts/time_large_k.py 256 1024 2048 4096 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_time_large_k_prof.txt
(timeout 300 python scripts/time_large_k.py 256 1024 2048 4096; IFD_LARGE_STEPS=501 timeout 300 python scripts/time_large_k.py 256 1024 2048; IFD_LARGE_SCAN=1 timeout 300 python scripts/time_large_k.py 256 1024 2048 4096) 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_time_large_k.txt
bash scripts/large_k_groups.sh > gpurun_out/r06_large_k_groups.txt 2>&1
cut -c1-300 gpurun_out/r06_bench.json
