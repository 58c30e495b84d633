#!/bin/bash
# Round 6: the whole GPU suite on the current build + the default bench line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r06_gpu_tests_second.log 2>&1
tail -6 gpurun_out/r06_gpu_tests_second.log
timeout 900 python bench.py > gpurun_out/r06_bench_second.json 2> gpurun_out/r06_bench_second.err
python - <<'P'
import json
d = json.loads([l for l in open('gpurun_out/r06_bench_second.json') if l.startswith('{')][-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["launch_shapes"])
print("f32_equivalent", {k: d["f32_equivalent"].get(k) for k in ("value", "ms_per_file")}, d["f32_equivalent"].get("roofline", {}).get("frac"))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
for k, v in d["extras"].items():
    print(k, {a: b for a, b in v.items() if a not in ("what",) and not isinstance(b, dict)} if isinstance(v, dict) else v)
P
