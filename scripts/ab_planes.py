"""What does the tap-gather traffic cost?  The optimiser launch of the bench workload three ways, same clouds, same arithmetic:
  A  every cloud reads its own copy of ONE plane set        (B x 1.5 MiB of distinct memory: taps from HBM / Infinity Cache)
  B  every cloud reads the SAME 1.5 MiB                     (ifd_opt_params.planes_shared: taps hit in L2)
  C  the bench workload itself (every cloud its own planes) for reference
A and B run identical instruction streams on identical values (all clouds see plane set 0, so the trajectories differ from C,
but not between A and B); their launch-time difference is the price of the gather traffic.
    python scripts/ab_planes.py [clouds]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import ifdefense_amd as I

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2468
r = I.Restorer(I.weights.pack_state_dict(I.weights.random_state_dict(0)), device="cuda:0")
x = torch.from_numpy(bench.synth_clouds(n)).cuda()
prep = r.prepare(x, r.sor(x), seed=1234)
planes = r.encode_inputs(prep["sel"], prep["t_per_cloud"])
lb = torch.full((n,), 192, dtype=torch.int32, device="cuda")
copies = planes[:1].expand(n, -1, -1, -1, -1).contiguous()


def run(pl, shared):
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = r.optimize_points(prep["init"], pl, rep_weight=500.0, iterations=500, loss_batch=lb, split=1, planes_shared=shared)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    return best, out


a_ms, a = run(copies, False)
b_ms, b = run(copies, True)
c_ms, _ = run(planes, False)
assert torch.equal(a, b), "A and B must be the same computation"
print("clouds %d: A own copies of plane set 0: %.1f ms | B one shared plane set (L2-resident taps): %.1f ms | delta %.2f %% | "
      "C bench planes: %.1f ms" % (n, a_ms, b_ms, 100.0 * (a_ms - b_ms) / a_ms, c_ms))
