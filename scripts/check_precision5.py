"""-DIFD_BF_DBG=256 build: the tile reports (sum of the sampled features, sum of fc_p's outputs, logit) in place of the gradient;
which of them is not reproducible run to run?"""
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
ms = []
for rep in range(16):
    o, (m, v, t) = r.optimize_points(init, planes, rep_weight=0.0, steps=1, normalize=False, precision="bf16x6", split=1, return_state=True)
    ms.append(m.cpu().numpy() * 10.0)
ms = np.stack(ms)
med = np.median(ms, axis=0)
for comp, name in enumerate((os.environ.get("N0","sum c"), os.environ.get("N1","sum fc_p"), os.environ.get("N2","logit"))):
    dev = ms[..., comp] != med[None, ..., comp]
    groups = sorted(set((int(run), int(c), int(p) // 16) for run, c, p in np.argwhere(dev)))
    print(name, ": deviating (run, cloud, 16-point group):", groups[:12], "n =", len(groups))
    for run, c, gi in groups[:4]:
        print("    ", ms[run, c, gi * 16:gi * 16 + 4, comp], "median", med[c, gi * 16:gi * 16 + 4, comp])
