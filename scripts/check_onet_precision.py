"""ONet-Opt split precision against the f32 kernel: gradient of one teacher-forced step, 10 free steps, determinism, launch time."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ifdefense_amd as I  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
r = I.OnetRestorer(I.weights.pack_state_dict(I.weights.onet_random_state_dict(0), "onet"), device="cuda:0")
x = torch.from_numpy(bench.synth_clouds(n)).cuda()
prep = r.prepare(x, r.sor(x), n_sel=300, seed=1234)
c = r.encode_inputs(prep["sel"], prep["t_per_cloud"])
init = prep["init"]
res = {}
for mode in ("f32", "bf16x6", "bf16x3"):
    p1, (m1, v1, _) = r.optimize_points(init[:8], c[:8], rep_weight=500.0, steps=1, return_state=True, normalize=False, precision=mode)
    p1b, (m1b, _, _) = r.optimize_points(init[:8], c[:8], rep_weight=500.0, steps=1, return_state=True, normalize=False, precision=mode)
    p10 = r.optimize_points(init[:8], c[:8], rep_weight=500.0, steps=10, normalize=False, precision=mode)
    res[mode] = (m1.cpu().numpy() / 0.1, p1.cpu().numpy(), p10.cpu().numpy(), bool(torch.equal(m1, m1b) and torch.equal(p1, p1b)))
g32 = res["f32"][0]
gmax = np.abs(g32).max()
for mode in ("bf16x6", "bf16x3"):
    g = res[mode][0]
    err = np.abs(g - g32).max(-1) / gmax
    d10 = np.linalg.norm(res[mode][2] - res["f32"][2], axis=-1)
    print("%-7s gradient vs the f32 kernel: median %.2e, 99th %.2e, max %.2e of max (points beyond 1e-5: %d of %d); x after 1 step max %.1e; "
          "10 free steps: max %.2e median %.2e; run-to-run identical: %s" %
          (mode, np.median(err), np.quantile(err, 0.99), err.max(), int((err > 1e-5).sum()), err.size,
           np.abs(res[mode][1] - res["f32"][1]).max(), d10.max(), np.median(d10), res[mode][3]))
for mode in ("f32", "bf16x6", "bf16x3"):
    r.optimize_points(init[:8], c[:8], rep_weight=500.0, steps=2, precision=mode)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r.optimize_points(init, c, rep_weight=500.0, steps=51, precision=mode)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    flop = n * 1024 * 51 * 2 * 2 * (10 * 256 * 256 + 4 * 256)
    print("%-7s %d clouds x 51 steps: %.1f ms -> %.1f clouds/s at 501 steps, %.1f f32-equivalent TFLOP/s" %
          (mode, n, dt * 1e3, n / (dt * 501 / 51), flop / dt / 1e12))
