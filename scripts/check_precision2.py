"""Determinism probes of the split-precision tiles: run to run, split vs unsplit, step by step."""
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
for prec in ("bf16x6",):
    for steps in (1, 2, 3, 10):
        for rw in (500.0, 0.0):
            outs = {}
            for split in (1, 1, 2, 2, 4):
                o, (m, v, t) = r.optimize_points(init, planes, rep_weight=rw, steps=steps, normalize=False, precision=prec, split=split, return_state=True)
                outs.setdefault(split, []).append((o.cpu().numpy(), m.cpu().numpy(), v.cpu().numpy()))
            a = outs[1][0]
            def diff(b):
                return [int((a[i] != b[i]).sum()) for i in range(3)]
            print(prec, "steps", steps, "rep", rw, "run-to-run S=1:", diff(outs[1][1]), " S=2 vs S=1:", diff(outs[2][0]), " S=2 run-to-run:",
                  [int((outs[2][0][i] != outs[2][1][i]).sum()) for i in range(3)], " S=4 vs S=1:", diff(outs[4][0]))
            if steps == 1 and rw == 0.0:
                bad = np.argwhere((a[1] != outs[2][0][1]).any(-1))
                print("   points whose first moment differs (cloud, point):", bad[:20].tolist(), "of", len(bad))
                for c, p in bad[:5]:
                    print("     ", a[1][c, p], outs[2][0][1][c, p])
