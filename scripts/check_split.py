"""Split clouds (ifd_opt_params.split = 2 / 4: S workgroups per cloud) against the one-workgroup kernel: bitwise equality of
points, Adam moments and losses on the bench workload, and the time of a launch that does not fill the GPU."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import ifdefense_amd as I

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
r = I.Restorer(I.weights.pack_state_dict(I.weights.random_state_dict(0)), device="cuda:0")
x = torch.from_numpy(bench.synth_clouds(n)).cuda()
prep = r.prepare(x, r.sor(x), seed=1234)
planes = r.encode_inputs(prep["sel"], prep["t_per_cloud"])
lb = torch.full((n,), 192, dtype=torch.int32, device="cuda")
res = {}
for split in (1, 2, 4, 0):
    for K in (1024, 777):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = r.optimize_points(prep["init"][:, :K].contiguous(), planes, rep_weight=500.0, steps=steps, loss_batch=lb,
                                return_state=True, return_loss=True, split=split)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res[(split, K)] = out
        ref = res[(1, K)]
        same = [bool(torch.equal(out[0], ref[0])), bool(torch.equal(out[1][0], ref[1][0])), bool(torch.equal(out[1][1], ref[1][1])),
                bool(torch.equal(out[2], ref[2]))]
        print("split %d K %4d: %.1f ms  points/m/v/loss bitwise equal to split=1: %s  finite %s" %
              (split, K, dt * 1e3, same, bool(torch.isfinite(out[0]).all())), flush=True)
