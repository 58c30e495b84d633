"""Micro-benchmark of ifd_optimize alone (random planes, synthetic sphere-ish points)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ifdefense_amd as I  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--clouds", type=int, default=512)
ap.add_argument("--steps", type=int, default=501)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--rep_weight", type=float, default=500.0)
a = ap.parse_args()

r = I.Restorer(I.weights.pack_state_dict(I.weights.random_state_dict(0)), device="cuda:0")
g = torch.Generator().manual_seed(0)
B = a.clouds
v = torch.randn(B, 1024, 3, generator=g)
pts = (0.4 * v / v.norm(dim=-1, keepdim=True) + 0.01 * torch.randn(B, 1024, 3, generator=g)).cuda()
planes = (torch.randn(B, 3, 64, 64, 32, generator=g) * 0.5).cuda()
for _ in range(1):
    r.optimize_points(pts[:8], planes[:8], rep_weight=a.rep_weight, steps=5)
torch.cuda.synchronize()
for _ in range(a.reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = r.optimize_points(pts, planes, rep_weight=a.rep_weight, steps=a.steps)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    cps = B / (ms / 1e3)
    print("clouds %d steps %d: %.1f ms  -> %.1f clouds/s  | %.1f us/step/wave-of-256 | dense %.1f TF/s (%.1f%% of 157.3)"
          % (B, a.steps, ms, cps, ms * 1e3 / a.steps / max(1, (B + 255) // 256),
             cps * 2 * 15488 * 2 * 1024 * a.steps / 1e12, cps * 2 * 15488 * 2 * 1024 * a.steps / 157.3e12 * 100))
assert torch.isfinite(out).all()
c = r.counters()
print('effective shader clock while cloud 0 ran: %.2f GHz (cycles %d over the last launch of %.1f ms; valid for <= 256 clouds) -> %.1f k shader cycles per step' % (c['cloud0_shader_cycles'] / (ms * 1e-3) / 1e9, c['cloud0_shader_cycles'], ms, c['cloud0_shader_cycles'] / a.steps / 1e3))
print('counters', c, '-> rebuilds per wave per run: %.1f' % (c['knn_rebuilds'] / (B * 8)))
