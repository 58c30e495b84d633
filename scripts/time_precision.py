"""Launch time of 256 clouds x 501 steps per precision mode, own planes vs one shared plane set (taps from L2)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ifdefense_amd as I
r = I.Restorer(I.weights.pack_state_dict(I.weights.random_state_dict(0)), device="cuda:0")
g = torch.Generator().manual_seed(0)
B = int(os.environ.get("CLOUDS", "256"))
v = torch.randn(B, 1024, 3, generator=g)
pts = (0.4 * v / v.norm(dim=-1, keepdim=True) + 0.01 * torch.randn(B, 1024, 3, generator=g)).cuda()
planes = (torch.randn(B, 3, 64, 64, 32, generator=g) * 0.5).cuda()
for prec in sys.argv[1:] or ["f32", "bf16x6", "bf16x3"]:
    for shared in (False, True):
        for rw in (500.0, 0.0):
            r.optimize_points(pts[:8], planes[:8], rep_weight=rw, steps=5, precision=prec)
            best = 1e9
            for _ in range(2):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r.optimize_points(pts, planes, rep_weight=rw, steps=501, precision=prec, planes_shared=shared, split=1, check=False)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            c = r.counters()
            print("%-7s planes %-6s rep_weight %5.0f: %6.1f ms  %6.1f k cycles/step  clock %.2f GHz" % (
                prec, "shared" if shared else "own", rw, best, c["cloud0_shader_cycles"] / 501 / 1e3, c["cloud0_shader_cycles"] / (best * 1e-3) / 1e9))
