#!/bin/bash
# What the chip's power management does under each kernel: samples rocm-smi (socket power, sclk) every ~0.3 s while the f32 probes, the
# bench launches (f32, bf16x6, bf16x3) and the ONet-Opt launch run, one after the other on the same box.
#   bash scripts/power_clock_probe.sh  -> gpurun_out/r05_power_clock.txt
R=$(pwd); mkdir -p gpurun_out
S=/tmp/smi_samples.txt; : > $S
( while true; do
    echo "T $(date +%s.%N) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk' | tr -s ' ' | tr '\n' '|')" >> $S
    sleep 0.2
  done ) &
SAMPLER=$!
mark() { echo "MARK $(date +%s.%N) $1" >> $S; }
{
mark idle; sleep 3
mark sustained_rate; ./scripts/f32_sustained_rate
mark structure_probe; ./scripts/f32_tile_structure_probe
mark bench_f32; python bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline | cut -c1-400
mark bench_bf16x6; python bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline --profile-precision bf16x6 | cut -c1-400
mark bench_bf16x3; python bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline --profile-precision bf16x3 | cut -c1-400
mark onet; python bench.py --workload onet-opt --steps 2 --warmup 1 --no-extras --no-cpu-baseline | cut -c1-400
mark end
} > gpurun_out/r05_power_clock_runs.txt 2>&1
kill $SAMPLER
python - <<'P' > gpurun_out/r05_power_clock.txt
import re
seg, cur = {}, None
order = []
for l in open('/tmp/smi_samples.txt'):
    if l.startswith('MARK'):
        cur = l.split()[2]; order.append(cur); seg[cur] = []
    elif l.startswith('T') and cur:
        p = re.search(r'Power[^|]*?:\s*([0-9.]+)', l)
        c = re.search(r'sclk[^|]*?\((\d+)Mhz\)', l, flags=re.I)
        if p or c:
            seg[cur].append((float(p.group(1)) if p else None, int(c.group(1)) if c else None))
print("rocm-smi samples while each workload ran (socket power W: median / max; sclk MHz: median / min / max; n samples)")
for k in order:
    v = seg[k]
    pw = sorted(x[0] for x in v if x[0] is not None); ck = sorted(x[1] for x in v if x[1] is not None)
    if not v:
        print("%-16s no samples" % k); continue
    print("%-16s power %s / %s   sclk %s / %s / %s   n = %d" % (k, pw[len(pw) // 2] if pw else None, pw[-1] if pw else None,
          ck[len(ck) // 2] if ck else None, ck[0] if ck else None, ck[-1] if ck else None, len(v)))
print()
print(open('gpurun_out/r05_power_clock_runs.txt').read())
print("first raw sample lines:")
print(''.join(open('/tmp/smi_samples.txt').readlines()[:6]))
P
cat gpurun_out/r05_power_clock.txt | head -70
