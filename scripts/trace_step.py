"""Timeline of ONE optimiser step of cloud 0 (the middle step), per wave, from a -DIFD_TRACE build of libifd.so:
    scripts/build_variant.sh trace "-DIFD_TRACE" && cp if-defense_amd/csrc/libifd_v_trace.so if-defense_amd/csrc/libifd.so
    python scripts/trace_step.py [clouds]
Prints, per wave, the cycles spent in each sub-phase of the step (relative to the earliest step start)."""
import os, sys
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
r.optimize_points(prep["init"], planes, rep_weight=500.0, iterations=500, loss_batch=lb, split=int(os.environ.get("IFD_SPLIT", "0")))
tr = r.wave_trace()
t0 = min(w[7] for w in tr if w[7])
names = ["build", "eval", "rep", "tiles_end", "barrier1", "adam", "T", "start"]
print("wave  start  build   eval    rep | tile ends ... | loop-end barrier1 adam-end   (k cycles since the earliest start)")
for wi, w in enumerate(tr):
    rel = lambda v: (v - t0) / 1e3 if v else float("nan")
    tiles = [rel(v) for v in w[8:20] if v]
    print("%d   %6.1f %6.1f %6.1f %6.1f | %s | %6.1f %6.1f %6.1f" % (wi, rel(w[7]), rel(w[0]), rel(w[1]), rel(w[2]),
          " ".join("%6.1f" % t for t in tiles), rel(w[3]), rel(w[4]), rel(w[5])))
print("inside the kNN phase (k cycles): unparked, flags read, list words here, front evaluated | after the mid-step barrier: Adam done, PIX written")
for wi, w in enumerate(tr):
    print("%d   %s" % (wi, " ".join("%6.1f" % ((v - t0) / 1e3) if v else "   nan" for v in w[24:30])))
