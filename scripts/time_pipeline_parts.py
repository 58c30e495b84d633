"""Optimiser time on the bench workload (real encoder planes, bench clouds), with and without the repulsion term."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import ifdefense_amd as I

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2468
if os.environ.get("IFD_WEIGHTS") == "trained":      # the converged-surface regime (tests/golden/train_trained_like.py)
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "trained_like_f16.npz"))
    r = I.Restorer(I.weights.pack_state_dict({k: z[k].astype(np.float32) for k in z.files}), device="cuda:0")
else:
    r = I.Restorer(I.weights.pack_state_dict(I.weights.random_state_dict(0)), device="cuda:0")
x = torch.from_numpy(bench.synth_clouds(n)).cuda()
keep = r.sor(x)
prep = r.prepare(x, keep, seed=1234)
planes = r.encode_inputs(prep["sel"], prep["t_per_cloud"])
lb = torch.full((n,), 192, dtype=torch.int32, device="cuda")
for rw in (500.0, 0.0, 500.0):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r.optimize_points(prep["init"], planes, rep_weight=rw, iterations=500, loss_batch=lb)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    c = r.counters()
    rounds = (n + 255) // 256
    print("rep_weight %5.0f: %.1f ms  %.1f clouds/s  %.1f us/step/round  rebuilds/cloud %.1f passes/rebuild %.2f brute %d ring-evals/wave-step %.3f"
          % (rw, dt * 1e3, n / dt, dt * 1e6 / 501 / rounds, c["knn_rebuilds"] / 8 / n,
             c["knn_passes"] / max(1, c["knn_rebuilds"]), c["knn_brute_scans"], c["knn_ring_evals"] / (8.0 * n * 501)))
    if any(c["prof_cycles"]):
        tot = c["cloud0_shader_cycles"]
        names = ("build", "eval", "rep", "tiles", "wait", "adam")
        print("   cloud 0, mean over 8 waves, %% of kernel cycles: " +
              "  ".join("%s %.1f" % (nm, 100.0 * v / 8 / tot) for nm, v in zip(names, c["prof_cycles"])))
        print("   cycles/step: cloud0 %.0f  mean cloud %.0f  max cloud %.0f   (launch wall per round-step at 2.3 GHz: %.0f)" %
              (tot / 501, c["prof_cycles"][7] / n / 501, c["prof_cycles"][6] / 501, dt * 2.3e9 / 501 / rounds))
        print("   max rebuilds in a cloud %d" % (c["prof_cycles"][5] >> 32))
    print("   refresh wave-steps/wave-step %.3f, lists built per cloud-step %.1f, exact evals/wave-step %.4f" %
          (c["knn_refresh_waves"] / (8.0 * n * 501), c["knn_lists_built"] / (n * 501.0), c["knn_exact_evals"] / (8.0 * n * 501)))
