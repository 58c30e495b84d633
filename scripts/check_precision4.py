"""Which sub-tiles deviate between repeated runs of one bf16x6 step (rep_weight 0)?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ifdefense_amd as I
from oracle import convonet_oracle as O
PL = ("xz", "xy", "yz")
g0 = np.load(os.path.join(ROOT, "tests", "golden", "convonet_golden.npz"))
planes = {pl: torch.from_numpy(g0["planes01"][:, i]) for i, pl in enumerate(PL)}
r = I.Restorer(I.weights.pack_state_dict(O.make_random_weights(0)), device="cuda:0")
init = torch.from_numpy(g0["init_points"][:2])
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x6"
ms = []
for rep in range(16):
    o, (m, v, t) = r.optimize_points(init, planes, rep_weight=0.0, steps=1, normalize=False, precision=prec, split=1, return_state=True)
    ms.append(m.cpu().numpy())
ms = np.stack(ms)            # [16, 2, 1024, 3]
ref_f32 = r.optimize_points(init, planes, rep_weight=0.0, steps=1, normalize=False, precision="f32", split=1, return_state=True)[1][0].cpu().numpy()
# majority value per point: the median over runs
med = np.median(ms, axis=0)
dev = np.abs(ms - med[None]).max(-1)         # [16, 2, 1024]
scale = np.abs(med).max()
for run in range(16):
    bad = np.argwhere(dev[run] > 0)
    groups = sorted(set((int(c), int(p) // 16) for c, p in bad))
    for c, gidx in groups:
        pts = [p for cc, p in bad if cc == c and p // 16 == gidx]
        d = dev[run, c, gidx * 16:(gidx + 1) * 16]
        e32 = np.abs(ms[run, c, gidx * 16:(gidx + 1) * 16] - ref_f32[c, gidx * 16:(gidx + 1) * 16]).max() / scale
        em = np.abs(med[c, gidx * 16:(gidx + 1) * 16] - ref_f32[c, gidx * 16:(gidx + 1) * 16]).max() / scale
        print("run %2d cloud %d tile %2d sub-tile %d: %2d points deviate from the median run, max %.2e of the gradient's max; vs f32: this run %.2e, median run %.2e"
              % (run, c, gidx // 2, gidx % 2, len(pts), d.max() / scale, e32, em))
print("median run vs f32: max %.2e of max" % (np.abs(med - ref_f32).max() / scale))
