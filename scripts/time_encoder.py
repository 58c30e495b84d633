"""Time of the pre-processing kernels of one 2468-cloud pass (SOR, prepare, encode_points, U-Net) - HIP events."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import ifdefense_amd as I

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2468
r = I.Restorer(I.weights.pack_state_dict(I.weights.random_state_dict(0)), device="cuda:0")
x = torch.from_numpy(bench.synth_clouds(n)).cuda()


def timed(f, reps=3):
    f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = f(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best, out


t_sor, keep = timed(lambda: r.sor(x))
t_prep, prep = timed(lambda: r.prepare(x, keep, seed=1234))
t_pts, pre = timed(lambda: r.encode_points(prep["sel"], prep["t_per_cloud"]))
t_unet, planes = timed(lambda: r.unet(pre))
flop = n * 3.66e9
print("clouds %d: sor %.2f ms | prepare %.2f ms | encode_points %.2f ms | unet %.2f ms (%.1f TFLOP/s, %.1f%% of 157.3) | total %.2f ms"
      % (n, t_sor, t_prep, t_pts, t_unet, flop / t_unet / 1e9, flop / t_unet / 1e9 / 157.3 * 100, t_sor + t_prep + t_pts + t_unet))
