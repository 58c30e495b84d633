"""Stream groups of the K > 1024 path (api.cpp large_optimize_in_groups): the same call repeated - the groups settle in a different
phase from run to run, the results must not move.  Digests of points / moments / losses over N repetitions, f32 and bf16x6, against a
ONE-group context (measurement hook)."""
import hashlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import ifdefense_amd as I

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
w = I.weights.pack_state_dict(I.weights.random_state_dict(0))
os.environ["IFD_ENABLE_TEST_HOOKS"] = "1"
os.environ["IFD_TEST_LARGE_GROUPS"] = "1"
one = I.Restorer(w, device="cuda:0")
del os.environ["IFD_TEST_LARGE_GROUPS"]
r = I.Restorer(w, device="cuda:0")


def digest(out):
    h = hashlib.sha256()
    for t in (out[0], out[1][0], out[1][1], out[2]):
        h.update(t.cpu().numpy().tobytes())
    return h.hexdigest()[:16]


for n, k in ((256, 2048), (100, 1500), (600, 1100)):
    x = torch.from_numpy(bench.synth_clouds(n)).cuda()
    prep = r.prepare(x, r.sor(x), n_sel=600, n_opt=k, seed=77)
    planes = r.encode_inputs(prep["sel"], prep["t_per_cloud"])
    for prec in ("f32", "bf16x6"):
        kw = dict(rep_weight=500.0, steps=201, precision=prec, return_state=True, return_loss=True)
        ref = digest(one.optimize_points(prep["init"], planes, **kw))
        got = [digest(r.optimize_points(prep["init"], planes, **kw)) for _ in range(reps)]
        same = all(g == ref for g in got)
        print("%4d clouds x %4d points x 201 steps, %-6s: one group %s; %d repetitions in groups: %s" % (
            n, k, prec, ref, reps, "all identical to it" if same else "DIFFER: %s" % sorted(set(got))))
        assert same
print("ok")
