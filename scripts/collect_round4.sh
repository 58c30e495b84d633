#!/bin/bash
# Everything profiles/r04_* comes from, in one gpurun call (run from the repo root on the GPU box; ~15 min):
#   variants needed (build container): scripts/build_variant.sh trace "-DIFD_TRACE"; trace2_1 / trace2_3 "-DIFD_TRACE -DIFD_TRACE2=<n>";
#   IFD_EXTRA_FLAGS=-DIFD_PROF python if-defense_amd/build.py --force && cp csrc/libifd.so csrc/libifd_prof.so && python if-defense_amd/build.py --force
mkdir -p gpurun_out
R=$(pwd)
bash scripts/collect_profiles.sh r04 > gpurun_out/collect_r04.log 2>&1
bash scripts/pmc_bench.sh r04 > gpurun_out/pmc_bench_r04.log 2>&1
python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err
python bench.py --no-overlap --no-extras --no-cpu-baseline > gpurun_out/r04_bench_serial.json 2>/dev/null
for n in 309 617; do python bench.py --clouds $n --no-extras --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/r04_strong_proxy_$n.json 2>/dev/null; done
# one-step wave timeline and the inside of one tile per wave (second and last tile of the step)
cp if-defense_amd/csrc/libifd_v_trace.so if-defense_amd/csrc/libifd_trace.so 2>/dev/null
bash scripts/run_trace.sh > gpurun_out/r04_trace.txt 2>&1
for v in 1 3; do IFD_LIB=$R/if-defense_amd/csrc/libifd_v_trace2_$v.so python scripts/tile_trace.py 256 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_tile_trace_$v.txt; done
bash scripts/run_prof.sh 2468 > gpurun_out/r04_prof.txt 2>&1
python scripts/check_split.py 53 501 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_check_split.txt
bash scripts/unet_layers.sh 2468 > gpurun_out/r04_unet_layers.txt 2>&1
# ONet-Mesh: per-kernel time, and the decoder kernel's own roofline fraction
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mesh_prof && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mesh_prof -o m -- python $R/scripts/time_mesh.py 64 > /tmp/mesh_prof.log 2>&1
cd $R
python - <<'P' > gpurun_out/r04_onet_mesh_kernel_stats.txt 2>&1
import csv, glob, re
log = open('/tmp/mesh_prof.log').read()
print(log.strip().splitlines()[-2] if log.strip() else 'no output')
print(log.strip().splitlines()[-1] if log.strip() else '')
f = glob.glob('/tmp/mesh_prof/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = {}
for r in rows:
    n = r['Kernel_Name'].split('(')[0][:60]
    tot.setdefault(n, [0, 0.0])
    tot[n][0] += 1; tot[n][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
allms = sum(v[1] for v in tot.values())
print("kernel, calls, total ms, share (whole script: warm-up + two timed mesh_sample calls of 64 clouds)")
for n, (c, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%-62s %5d %9.2f %6.1f %%" % (n, c, ms, 100 * ms / allms))
# grid points of the two timed calls from the script's lines
pts = [float(x) for x in re.findall(r"\| (\d+) grid points/cloud", log)]
ge = tot.get([k for k in tot if 'onet_grid_eval' in k][0]) if any('onet_grid_eval' in k for k in tot) else None
if ge and pts:
    # the warm-up call (4 clouds) evaluates ~pts[0] points per cloud as well
    npts = 64 * sum(pts) + 4 * pts[0]
    flop = npts * 2 * (10 * 256 * 256 + 4 * 256)
    print("onet_grid_eval_kernel: %.3g grid points, %.2f ms in the kernel -> %.1f TFLOP/s = %.3f of the f32-MFMA peak (157.3)" %
          (npts, ge[1], flop / ge[1] / 1e9, flop / ge[1] / 1e9 / 157.3))
P
cut -c1-300 gpurun_out/r04_bench.json
