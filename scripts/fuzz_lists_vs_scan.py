"""Stress test: the certified-neighbour-list path must equal the exact scan bit for bit, for many shapes / sizes /
step counts (this is how the exact-tie instability of the scan's insertion was found).  Usage: python scripts/fuzz_lists_vs_scan.py [rounds]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ifdefense_amd as I  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
r = I.Restorer(I.weights.pack_state_dict(I.weights.random_state_dict(0)), device="cuda:0")
rng = np.random.default_rng(0)
bad = 0
for it in range(rounds):
    n = 28
    clouds = bench.synth_clouds(n, seed=100 + it)
    kind = it % 3
    if kind == 1:
        clouds = bench.knn_attack_like(clouds, seed=it)
    elif kind == 2:
        clouds = bench.subsample_like(clouds, 256, seed=it)
    x = torch.from_numpy(clouds).cuda()
    keep = r.sor(x)
    prep = r.prepare(x, keep, seed=it)
    planes = r.encode_inputs(prep["sel"], prep["t_per_cloud"])
    for K in (1024, int(rng.integers(6, 1024)), int(rng.integers(6, 200))):
        steps = int(rng.choice([60, 200, 501]))
        rw = float(rng.choice([500.0, 50.0, 2000.0]))
        init = prep["init"][:, :K].contiguous()
        a = r.optimize_points(init, planes, rep_weight=rw, steps=steps, normalize=False)
        c = r.counters()
        b = r.optimize_points(init, planes, rep_weight=rw, steps=steps, normalize=False, knn_scan_every_step=True)
        ok = torch.equal(a, b)
        bad += 0 if ok else 1
        print("round %d kind %d K %4d steps %3d rep_weight %6.0f: %s  (rebuilds/cloud %.1f, scans %d, refresh wave-steps %d)" %
              (it, kind, K, steps, rw, "equal" if ok else "DIFFERENT in clouds %s" % ((a - b).abs().amax((1, 2)) > 0).nonzero().flatten().tolist(),
               c["knn_rebuilds"] / 8 / n, c["knn_brute_scans"], c["knn_refresh_waves"]))
print("FAILURES:", bad)
sys.exit(1 if bad else 0)
