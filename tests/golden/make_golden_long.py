#!/usr/bin/env python
"""Late-trajectory fixtures (Adam state -> next state at t in {100, 500}) from the REFERENCE's
modules: the same harness as make_golden.py (imported for its shims, model builder and optimize loop), run for 500
steps on the two fixture clouds, recording (x, g, m, v, x_next, losses) at steps 99 and 499 (0-based; Adam's
t = 100 and 500).  Writes tests/golden/convonet_golden_long.npz.  Needs /root/reference; build container only.

    python tests/golden/make_golden_long.py
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
    model = MG.build_reference_model()
    planes2 = {pl: torch.from_numpy(g["planes01"][:, i]) for i, pl in enumerate(("xz", "xy", "yz"))}
    init = torch.from_numpy(g["init_points"][:2])
    rec = (99, 499)
    ns = MG.RD.load(model, MG.RD.default_args())
    _, snaps = MG.RD.run_optimize(ns, init, planes2, iterations=499, record=rec)     # the reference's own optimize_points
    out = {}
    for i in rec:
        for k in ("x", "g", "m", "v", "x_next"):
            out[f"traj{i}_{k}"] = snaps[i][k].numpy()
        out[f"traj{i}_loss"] = np.array([snaps[i]["occ"], snaps[i]["rep"]], np.float64)
    path = os.path.join(HERE, "convonet_golden_long.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
