"""MFMA point encoder against the thread-per-point one (env IFD_ENC_VALU=1, read once per process): the pre-U-Net planes and
the per-point features c of the same bench clouds through both, difference relative to the maximum, and the time of each.
    python scripts/enc_check.py [clouds]"""
import os, subprocess, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
if len(sys.argv) > 2:                                    # child: run one variant, save planes + c
    import torch
    sys.path.insert(0, root)
    import bench
    import ifdefense_amd as I
    r = I.Restorer(I.weights.pack_state_dict(I.weights.random_state_dict(0)), device="cuda:0")
    x = torch.from_numpy(bench.synth_clouds(n)).cuda()
    keep = r.sor(x)
    prep = r.prepare(x, keep, seed=1234)
    pre = r.encode_points(prep["sel"], prep["t_per_cloud"])
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); pre = r.encode_points(prep["sel"], prep["t_per_cloud"]); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    np.save(sys.argv[2], pre[:16].cpu().numpy())
    print("%s: encode_points %.2f ms for %d clouds" % (sys.argv[2], best, n))
    sys.exit(0)
outs = []
for tag, env in (("valu", "1"), ("mfma", "0")):
    f = "/tmp/enc_%s.npy" % tag
    e = dict(os.environ, IFD_ENC_VALU=env)
    r = subprocess.run([sys.executable, __file__, str(n), f], env=e, capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-2000:])
    outs.append(np.load(f))
d, w = outs
print("max |mfma - valu| / max |valu| = %.3e   (max |valu| %.3f, occupied cells equal %s, any nan %s)"
      % (np.abs(w - d).max() / np.abs(d).max(), np.abs(d).max(), bool(((w != 0) == (d != 0)).all()), bool(np.isnan(w).any())))
