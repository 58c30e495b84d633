"""Micro-benchmark of ifd_onet_optimize (ONet-Opt decoder variant): clouds/s and MFMA fraction."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ifdefense_amd as I  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 51
r = I.OnetRestorer(I.weights.pack_state_dict(I.weights.onet_random_state_dict(0), "onet"), device="cuda:0")
x = torch.from_numpy(bench.synth_clouds(n)).cuda()
keep = r.sor(x)
prep = r.prepare(x, keep, n_sel=300, seed=1234)
torch.cuda.synchronize(); t0 = time.perf_counter()
c = r.encode_inputs(prep["sel"], prep["t_per_cloud"])
torch.cuda.synchronize(); t_enc = time.perf_counter() - t0
r.optimize_points(prep["init"][:8], c[:8], rep_weight=500.0, steps=2)
for rw in (500.0, 0.0):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = r.optimize_points(prep["init"], c, rep_weight=rw, steps=steps)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    flop = n * 1024 * steps * 2 * 2 * (10 * 256 * 256 + 4 * 256)          # fwd + input-bwd
    print("rep_weight %3.0f: %d clouds x %d steps: %.1f ms -> %.2f ms/step/round, %.1f TFLOP/s (%.1f%% of 157.3), "
          "%.1f clouds/s at 501 steps | encoder %.1f ms" %
          (rw, n, steps, dt * 1e3, dt * 1e3 / steps / ((n + 255) // 256), flop / dt / 1e12, flop / dt / 157.3e10,
           n / (dt * 501 / steps), t_enc * 1e3))
assert torch.isfinite(out).all()
