"""Inside ONE decoder tile per wave (the n-th tile each wave runs in the middle step of cloud 0), region by region, from a
-DIFD_TRACE -DIFD_TRACE2=<n> build (the substitute for rocprofv3 --att / PC sampling, which the GPU boxes do not offer:
profiles/r04_att_pcsamp_unavailable.txt):
    scripts/build_variant.sh trace2_1 "-DIFD_TRACE -DIFD_TRACE2=1"
    IFD_LIB=$PWD/if-defense_amd/csrc/libifd_v_trace2_1.so python scripts/tile_trace.py [clouds]
Prints per wave the duration of every section of the tile, the 16-MFMA regions of the MLP against their 512-cycle MFMA
floor, and - for the two waves of each SIMD - how their tiles overlap."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import ifdefense_amd as I

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
r = I.Restorer(I.weights.pack_state_dict(I.weights.random_state_dict(0)), device="cuda:0")
x = torch.from_numpy(bench.synth_clouds(n)).cuda()
prep = r.prepare(x, r.sor(x), seed=1234)
planes = r.encode_inputs(prep["sel"], prep["t_per_cloud"])
lb = torch.full((n,), 192, dtype=torch.int32, device="cuda")
r.optimize_points(prep["init"], planes, rep_weight=500.0, iterations=500, loss_batch=lb, split=1)
tr = np.array(r.tile_trace(), dtype=np.int64)          # [8][128]
wt = r.wave_trace()
t0 = min(int(w[7]) for w in wt if w[7])
SEC = [("gather issue", 0, 1), ("tap wait", 1, 2), ("sample fwd", 2, 3), ("fc_p + first weights", 3, 4), ("fwd MLP (30 regions)", 4, 34),
       ("logit / seed", 34, 35), ("bwd MLP (30 regions)", 35, 65), ("fc_p bwd", 65, 66), ("re-gather issue", 66, 67),
       ("re-gather wait", 67, 68), ("sample bwd (+plane 3 wait)", 68, 69), ("lane reduce", 69, 70)]
print("tile traced per wave (index in the step's tile queue):", [int(t[71]) for t in tr])
print("\nsections, cycles per wave (waves 0-7; SIMD s holds waves s and s + 4)")
print("%-28s" % "section" + "".join("%8d" % w for w in range(8)) + "     mean")
tot = np.zeros(8)
for name, a, b in SEC:
    d = (tr[:, b] - tr[:, a]).astype(float)
    d[(tr[:, b] == 0) | (tr[:, a] == 0)] = np.nan
    tot += np.nan_to_num(d)
    print("%-28s" % name + "".join("%8.0f" % v for v in d) + "  %7.0f" % np.nanmean(d))
print("%-28s" % "tile" + "".join("%8.0f" % v for v in (tr[:, 70] - tr[:, 0])) + "  %7.0f" % np.mean(tr[:, 70] - tr[:, 0]))
print("\nMLP regions (16 MFMAs = 512 cycles of the matrix pipe each), cycles, mean over the 8 waves | min | max")
names = ["R1", "R2", "R3", "R4", "R5", "R6"]
for half, base in (("fwd", 4), ("bwd", 35)):
    print(half + "  " + " ".join("blk%d:" % i + "".join("%5s" % nm for nm in names) for i in range(1)))
    for i in range(5):
        d = np.array([[tr[w, base + 6 * i + k + 1] - tr[w, base + 6 * i + k] for k in range(6)] for w in range(8)], dtype=float)
        print("  block %d  mean " % i + "".join("%6.0f" % v for v in d.mean(0)) + "   min " + "".join("%6.0f" % v for v in d.min(0)) +
              "   max " + "".join("%6.0f" % v for v in d.max(0)))
reg = np.concatenate([np.diff(tr[:, 4:35], axis=1), np.diff(tr[:, 35:66], axis=1)], axis=1).astype(float)
print("all 60 regions: mean %.0f cycles (floor 512 when the wave has the pipe to itself, 1024 when its SIMD partner streams too); "
      "median %.0f, p90 %.0f" % (reg.mean(), np.median(reg), np.percentile(reg, 90)))
print("\nwhere each traced tile sits in the step (k cycles since the step's first wave started) and what the SIMD partner did meanwhile")
for w in range(8):
    p = (w + 4) % 8
    a, b = tr[w, 0], tr[w, 70]
    pa, pb = tr[p, 0], tr[p, 70]
    ov = max(0, min(b, pb) - max(a, pa))
    ends = [int(v) for v in wt[p][8:20] if v]
    print("wave %d: tile %5.1f .. %5.1f k (%.1f k) | partner wave %d traced tile %5.1f .. %5.1f k, overlap %.1f k; partner's tile ends at %s"
          % (w, (a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, p, (pa - t0) / 1e3, (pb - t0) / 1e3, ov / 1e3,
             " ".join("%.1f" % ((e - t0) / 1e3) for e in ends)))
# matrix-pipe accounting over the window in which BOTH waves of a SIMD are inside their traced tiles' MLP sections
print("\nSIMD accounting while both waves of a SIMD are inside the MLP sections of their traced tiles:")
for sd in range(4):
    w, p = sd, sd + 4
    lo = max(tr[w, 4], tr[p, 4]); hi = min(tr[w, 65], tr[p, 65])
    if hi <= lo:
        print("SIMD %d: the two traced tiles' MLP sections do not overlap" % sd); continue
    def regions_in(wv):
        k = 0.0
        for a, b in list(zip(tr[wv, 4:34], tr[wv, 5:35])) + list(zip(tr[wv, 35:65], tr[wv, 36:66])):
            o = max(0, min(b, hi) - max(a, lo))
            k += o / max(1, b - a)
        return k
    nr = regions_in(w) + regions_in(p)
    print("SIMD %d: window %.1f k cycles, %.1f regions of 16 MFMAs completed by the two waves -> %.1f k MFMA cycles = %.1f %% of the window"
          % (sd, (hi - lo) / 1e3, nr, nr * 0.512, 100.0 * nr * 512 / (hi - lo)))
