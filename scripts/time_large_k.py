"""ifd_optimize on clouds of more than 1024 points (the two-launch-per-step path): time per point and step against the persistent kernel."""
import hashlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ifdefense_amd as I  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ks = [int(a) for a in sys.argv[2:]] or [1024, 2048]
r = I.Restorer(I.weights.pack_state_dict(I.weights.random_state_dict(0)), device="cuda:0")
x = torch.from_numpy(bench.synth_clouds(n)).cuda()
base = None
for k in ks:
    steps = int(os.environ.get("IFD_LARGE_STEPS", "101"))           # (the first step of a launch builds the lists: time enough steps)
    scan = os.environ.get("IFD_LARGE_SCAN", "0") == "1"
    prec = os.environ.get("IFD_LARGE_PRECISION", "f32")
    prep = r.prepare(x, r.sor(x), n_sel=600, n_opt=k, seed=1234)
    planes = r.encode_inputs(prep["sel"], prep["t_per_cloud"])
    r.optimize_points(prep["init"][:8], planes[:8], rep_weight=500.0, steps=2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = r.optimize_points(prep["init"], planes, rep_weight=500.0, steps=steps, knn_scan_every_step=scan, precision=prec)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    digest = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:12]
    c = r.counters()
    per = dt / (n * k * steps)
    base = base or per
    print("K = %5d: %d clouds x %d steps in %.1f ms -> %.2f ns per point and step (%.2f x the first line) [points %s]%s" % (
        k, n, steps, dt * 1e3, per * 1e9, per / base, digest,
        "" if k <= 1024 else "; %.1f list epochs per cloud, %.2f %% of the point-steps through the exact query" %
        (c["knn_rebuilds"] / float(n), 100.0 * c["knn_exact_evals"] / (float(n) * k * steps))))
    if k > 1024 and c["prof_cycles"][3]:          # a -DIFD_PROF build: cloud 0's phase cycles per (non-first) step
        pc = c["prof_cycles"]
        print("        cloud 0, k cycles per step: list evaluation + certified terms %.1f | queued points %.1f | Adam %.1f" %
              (pc[0] / pc[3] / 1e3, pc[1] / pc[3] / 1e3, pc[2] / pc[3] / 1e3))
        if pc[7]:
            print("        cloud 0, queued points by reason: keys 5 / 6 tie %d, no valid list (rho = 0) %d, certificate %d" %
                  (pc[7] & 0xfffff, (pc[7] >> 20) & 0xfffff, pc[7] >> 40))
        if pc[6]:
            print("        wave 0 of cloud 0: %.2f passes of the queued-point loop per step, %.1f k cycles per pass in the candidate loop, %.1f k in fill + merge"
                  % (pc[6] / float(pc[3]), pc[4] / pc[6] / 1e3, pc[5] / pc[6] / 1e3))
