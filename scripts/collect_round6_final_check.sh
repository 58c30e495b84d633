#!/bin/bash
# what the driver runs at round end, on the final tree: pytest -m gpu, smoke(), the default bench line
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gpu_tests_final.log 2>&1
tail -3 gpurun_out/r06_gpu_tests_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err
python - <<'P'
import json
d = json.loads([l for l in open('gpurun_out/r06_bench_final.json') if l.startswith('{')][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["launch_shapes"])
fe = d["f32_equivalent"]; print(fe["value"], fe["roofline"]["frac"], fe["roofline"]["traffic"])
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["extras"]["k2048"]["per_point_cost_vs_1024"], d["extras"]["trained_like_full"]["value"])
P
