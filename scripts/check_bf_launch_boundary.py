"""Round 6 debugging aid: in bf16x6 the config #5 (K = 824, no SOR) attribution test found 'one launch of 10 steps' != 'ten launches of one
step with the Adam state carried' (last-bit differences).  Which of the two is unstable, from which step on, and does it depend on
the split?    python scripts/check_bf_launch_boundary.py [mode]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import ifdefense_amd as I

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x6"
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "convonet_golden.npz"))
r = I.Restorer(I.weights.pack_state_dict(I.weights.random_state_dict(0)), device="cuda:0")
for name, clouds, sor in (("K=824 noSOR", bench.drop_like(g["raw"]), False), ("K=824 SOR", bench.drop_like(g["raw"]), True),
                          ("K=256 noSOR", bench.subsample_like(g["raw"], 256), False)):
    x = torch.from_numpy(clouds).cuda()
    prep = r.prepare(x, r.sor(x) if sor else None, seed=11)
    planes = r.encode_inputs(prep["sel"], prep["t_per_cloud"])
    init = prep["init"]
    B = init.shape[0]
    for split in (0, 1, 2, 4):
        def one(n):
            return r.optimize_points(init, planes, rep_weight=500.0, steps=n, loss_batch=B, normalize=False, precision=mode, split=split)

        def stepwise(n):
            xx, st, outs = init, None, []
            for _ in range(n):
                xx, st = r.optimize_points(xx, planes, rep_weight=500.0, steps=1, loss_batch=B, normalize=False, state=st, return_state=True,
                                           precision=mode, split=split)
                outs.append(xx.clone())
            return outs

        a = [one(10) for _ in range(3)]
        s1, s2 = stepwise(10), stepwise(10)
        first = None
        for k in range(10):
            if not torch.equal(one(k + 1), s1[k]):
                first = k + 1
                break
        print("%s %s split=%d: one-launch runs equal %s | stepwise runs equal %s | one launch == stepwise %s (first differing step: %s; "
              "max |diff| after 10 steps %.2e)" % (mode, name, split, torch.equal(a[0], a[1]) and torch.equal(a[0], a[2]),
                                                  all(torch.equal(p, q) for p, q in zip(s1, s2)), torch.equal(a[0], s1[-1]), first,
                                                  float((a[0] - s1[-1]).abs().max())))
