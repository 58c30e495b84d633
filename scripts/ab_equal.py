"""Are two builds of libifd.so bit-identical on the bench workload?  python scripts/ab_equal.py libA.so libB.so [clouds] [steps]
Each library runs in its own process (IFD_LIB); compares the optimised points, both Adam moments and the reported losses."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch, bench, ifdefense_amd as I
    n, steps, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    r = I.Restorer(I.weights.pack_state_dict(I.weights.random_state_dict(0)), device="cuda:0")
    x = torch.from_numpy(bench.synth_clouds(n)).cuda()
    prep = r.prepare(x, r.sor(x), seed=1234)
    planes = r.encode_inputs(prep["sel"], prep["t_per_cloud"])
    lb = torch.full((n,), 192, dtype=torch.int32, device="cuda")
    p, (m, v, _), loss = r.optimize_points(prep["init"], planes, rep_weight=500.0, steps=steps, loss_batch=lb, normalize=False,
                                           return_state=True, return_loss=True)
    np.savez(out, p=p.cpu().numpy(), m=m.cpu().numpy(), v=v.cpu().numpy(), loss=loss.cpu().numpy())
    sys.exit(0)
a, b = sys.argv[1], sys.argv[2]
n = sys.argv[3] if len(sys.argv) > 3 else "300"
steps = sys.argv[4] if len(sys.argv) > 4 else "120"
res = []
for k, lib in enumerate((a, b)):
    out = "/tmp/ab_equal_%d.npz" % k
    subprocess.run([sys.executable, __file__, "--child", n, steps, out], check=True, env=dict(os.environ, IFD_LIB=os.path.abspath(lib)))
    res.append(np.load(out))
for key in ("p", "m", "v", "loss"):
    eq = np.array_equal(res[0][key], res[1][key])
    print("%-5s %s  (max |diff| %.3e)" % (key, "bit-identical" if eq else "DIFFERENT", float(np.abs(res[0][key] - res[1][key]).max())))
