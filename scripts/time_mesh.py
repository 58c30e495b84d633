"""Throughput of the ONet-Mesh path (encode -> MISE grid -> marching cubes -> surface samples) on synthetic clouds."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ifdefense_amd as I  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
r = I.OnetRestorer(I.weights.pack_state_dict(I.weights.onet_random_state_dict(0), "onet"), device="cuda:0")
x = torch.from_numpy(bench.synth_clouds(n)).cuda()
keep = r.sor(x)
prep = r.prepare(x, keep, n_sel=300, seed=1)
c = r.encode_inputs(prep["sel"], prep["t_per_cloud"])
g = torch.Generator().manual_seed(9)
med = float(r.decode((torch.rand(8, 4096, 3, generator=g) - 0.5) * 1.1, c[:8]).median())
thr = 1.0 / (1.0 + np.exp(-med))            # random weights: cut the field at its median so that there is a surface
r.mesh_sample(c[:4], threshold=thr)
for t, name in ((thr, "median-cut"), (None, "cfg 0.2")):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = r.mesh_sample(c, threshold=t)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    k = r.counters()
    flop = k["mesh_points"] * 2 * (10 * 256 * 256 + 4 * 256)
    print("%-10s: %d clouds in %.1f ms -> %.1f clouds/s | %.0f grid points/cloud in %d rounds, %.0f triangles/cloud, "
          "decoder %.1f TFLOP/s if it were all of the time" %
          (name, n, dt * 1e3, n / dt, k["mesh_points"] / n, k["mesh_rounds"], float(out["n_triangles"].float().mean()),
           flop / dt / 1e12))
