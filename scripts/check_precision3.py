"""Run-to-run determinism of one bf16x6 step (rep_weight 0) for the library in IFD_LIB."""
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
ref = None
nd = 0
for rep in range(int(os.environ.get("REPS", "12"))):
    o, (m, v, t) = r.optimize_points(init, planes, rep_weight=float(os.environ.get("RW", "0")), steps=int(os.environ.get("STEPS", "1")), normalize=False, precision=prec, split=1, return_state=True)
    m = m.cpu().numpy()
    if ref is None: ref = m
    else:
        d = int((m != ref).any(-1).sum())
        nd += d
print(os.environ.get("IFD_LIB", "default"), prec, "points whose gradient differs from run 0, summed over 11 repeats:", nd)
