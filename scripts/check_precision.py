"""Split-precision decoder tiles (ifd_opt_params.precision; tile_bf.h) against the reference fixtures and against each other:
the hot tile's gradient recovered from Adam's first moment (tests/test_gpu_parity.py::_hot_gradient) for f32 / bf16x6 / bf16x3,
ten free-running steps, and the launch time of 256 clouds x 501 steps on synthetic planes (scripts/time_optimize.py's workload)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ifdefense_amd as I  # noqa: E402
from oracle import convonet_oracle as O  # noqa: E402

PL = ("xz", "xy", "yz")
g0 = np.load(os.path.join(ROOT, "tests", "golden", "convonet_golden.npz"))
gl = np.load(os.path.join(ROOT, "tests", "golden", "convonet_golden_long.npz"))
planes = {pl: torch.from_numpy(g0["planes01"][:, i]) for i, pl in enumerate(PL)}
r = I.Restorer(I.weights.pack_state_dict(O.make_random_weights(0)), device="cuda:0")
for prec in ("f32", "bf16x6", "bf16x3"):
    for f, steps in ((g0, (0, 1, 9, 49)), (gl, (99, 499))):
        for t in steps:
            x = torch.from_numpy(f[f"traj{t}_x"])
            m0, v0 = f[f"traj{t}_m"], f[f"traj{t}_v"]
            out, (m1, v1, t1) = r.optimize_points(x, planes, rep_weight=500.0, steps=1, normalize=False,
                                                  state=(torch.from_numpy(m0), torch.from_numpy(v0), t), return_state=True, precision=prec)
            g = m0 + (m1.cpu().numpy().astype(np.float64) - m0) / 0.1
            g_ref = f[f"traj{t}_g"].astype(np.float64)
            eg = np.abs(g - g_ref).max() / np.abs(g_ref).max()
            flips = int((np.abs(out.cpu().numpy() - f[f"traj{t}_x_next"]) > 1e-6).sum())
            print("%-7s t=%3d: gradient error %.2e of max, coordinates off by > 1e-6: %d" % (prec, t + 1, eg, flips))
    init = torch.from_numpy(g0["init_points"][:2])
    x10 = r.optimize_points(init, planes, rep_weight=500.0, steps=10, normalize=False, precision=prec)
    d10 = np.linalg.norm(x10.cpu().numpy() - g0["traj9_x_next"], axis=-1)
    print("%-7s 10 free steps: max per-point L2 %.2e median %.2e" % (prec, d10.max(), np.median(d10)))
    for split in (2, 4):
        xs = r.optimize_points(init, planes, rep_weight=500.0, steps=10, normalize=False, precision=prec, split=split)
        print("%-7s split %d bit-identical to one workgroup per cloud: %s" % (prec, split, bool((xs == x10).all())))

gen = torch.Generator().manual_seed(0)
B = 256
v = torch.randn(B, 1024, 3, generator=gen)
pts = (0.4 * v / v.norm(dim=-1, keepdim=True) + 0.01 * torch.randn(B, 1024, 3, generator=gen)).cuda()
pl = (torch.randn(B, 3, 64, 64, 32, generator=gen) * 0.5).cuda()
outs = {}
for prec in ("f32", "bf16x6", "bf16x3"):
    r.optimize_points(pts[:8], pl[:8], rep_weight=500.0, steps=5, precision=prec)
    for rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = r.optimize_points(pts, pl, rep_weight=500.0, steps=501, precision=prec)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    c = r.counters()
    outs[prec] = out.cpu().numpy()
    print("%-7s 256 clouds x 501 steps: %.1f ms -> %.0f clouds/s, %.1f k shader cycles per step, finite %s" %
          (prec, ms, B / ms * 1e3, c["cloud0_shader_cycles"] / 501 / 1e3, bool(torch.isfinite(out).all())))
for prec in ("bf16x6", "bf16x3"):
    d = np.linalg.norm(outs[prec] - outs["f32"], axis=-1)
    print("%-7s vs f32 after 501 free steps (chaotic): median per-point L2 %.2e, fraction > 1e-3 %.3f" % (prec, np.median(d), (d > 1e-3).mean()))
