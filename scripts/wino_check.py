"""Winograd F(2x2, 3x3) U-Net against the implicit-GEMM one (env IFD_UNET_DIRECT=1, read once per process): the same
pre-U-Net planes of the bench clouds through both, difference relative to the output's maximum, and the time of each.
    python scripts/wino_check.py [clouds]"""
import os, subprocess, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
if len(sys.argv) > 2:                                    # child: run one variant, save the planes
    import torch
    sys.path.insert(0, root)
    import bench
    import ifdefense_amd as I
    r = I.Restorer(I.weights.pack_state_dict(I.weights.random_state_dict(0)), device="cuda:0")
    x = torch.from_numpy(bench.synth_clouds(n)).cuda()
    keep = r.sor(x)
    prep = r.prepare(x, keep, seed=1234)
    pre = r.encode_points(prep["sel"], prep["t_per_cloud"])
    out = r.unet(pre)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = r.unet(pre); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    np.save(sys.argv[2], out[:16].cpu().numpy())
    print("%s: unet %.2f ms for %d clouds" % (sys.argv[2], best, n))
    sys.exit(0)
outs = []
for tag, env in (("direct", "1"), ("winograd", "0")):
    f = "/tmp/wino_%s.npy" % tag
    e = dict(os.environ, IFD_UNET_DIRECT=env)
    r = subprocess.run([sys.executable, __file__, str(n), f], env=e, capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-2000:])
    outs.append(np.load(f))
d, w = outs
print("max |winograd - direct| / max |direct| = %.3e   (max |direct| %.3f, any nan %s)"
      % (np.abs(w - d).max() / np.abs(d).max(), np.abs(d).max(), bool(np.isnan(w).any())))
