#!/usr/bin/env python
"""Second parity base (round 2): weight seed 1, FOUR clouds, trajectory snapshots of the REFERENCE's modules.

The round-1 fixtures pin the hot loop on one weight seed and two clouds.  This script drives the same harness
(make_golden.py: shims, reference model builder, the optimize_points loop around the reference's decode / repulsion
loss / torch.optim.Adam) with weights of seed 1 on all four fixture clouds (B = 4, so the 1/B loss factor differs
from the seed-0 fixtures too) and records (x, g, m, v, x_next, losses) at steps 0, 1, 9, 49, 99 plus the planes the
reference encoder produced.  Writes tests/golden/convonet_golden_seed1.npz (planes as float16-free float32, compressed).
Needs /root/reference; build container only.

    python tests/golden/make_golden_seed1.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (registers the shims, imports the reference modules)


def main():
    g = np.load(os.path.join(HERE, "convonet_golden.npz"))
    MG.WEIGHT_SEED = 1
    model = MG.build_reference_model()
    proc = [g["proc_pad"][b, :g["proc_len"][b]] for b in range(4)]
    sel = torch.from_numpy(np.stack([proc[b][g["sel_idx"][b]] for b in range(4)]))
    with torch.no_grad():
        planes = model.encode_inputs(sel)
    init = torch.from_numpy(g["init_points"])                       # [4,1024,3]
    rec = (0, 1, 9, 49, 99)
    ns = MG.RD.load(model, MG.RD.default_args())
    _, snaps = MG.RD.run_optimize(ns, init, planes, iterations=99, record=rec)       # the reference's own optimize_points
    out = {"planes": np.stack([planes[pl].numpy() for pl in ("xz", "xy", "yz")], 1)}      # [4,3,32,64,64]
    for i in rec:
        for k in ("x", "g", "m", "v", "x_next"):
            out[f"traj{i}_{k}"] = snaps[i][k].numpy()
        out[f"traj{i}_loss"] = np.array([snaps[i]["occ"], snaps[i]["rep"]], np.float64)
    # decoder logits and d(sum logits)/dp on the four init clouds (second seed of G2)
    p = init.clone().requires_grad_()
    logits = model.decode(p, planes).logits
    logits.sum().backward()
    out["dec_logits"] = logits.detach().numpy()
    out["dec_dlogit_dp"] = p.grad.numpy().copy()
    path = os.path.join(HERE, "convonet_golden_seed1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.2f MB" % (os.path.getsize(path) / 1e6))


if __name__ == "__main__":
    main()
