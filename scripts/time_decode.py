"""Micro-benchmark of the decoder tile alone (ifd_decode with input gradient): same device function as the
optimiser's phase A, no barriers / kNN / Adam."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ifdefense_amd as I  # noqa: E402


ap = argparse.ArgumentParser()
ap.add_argument("--clouds", type=int, default=2560)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--nograd", action="store_true")
a = ap.parse_args()
r = I.Restorer(I.weights.pack_state_dict(I.weights.random_state_dict(0)), device="cuda:0")
g = torch.Generator().manual_seed(0)
B = a.clouds
nb = min(B, 256)
v = torch.randn(nb, 1024, 3, generator=g)
pts = (0.4 * v / v.norm(dim=-1, keepdim=True)).repeat((B + nb - 1) // nb, 1, 1)[:B].contiguous().cuda()
planes = (torch.randn(nb, 3, 64, 64, 32, generator=g) * 0.5).cuda()
planes = planes.repeat((B + nb - 1) // nb, 1, 1, 1, 1)[:B].contiguous()
r.decode(pts[:8], planes[:8], want_grad=not a.nograd)
torch.cuda.synchronize()
for _ in range(a.reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r.decode(pts, planes, want_grad=not a.nograd)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    rounds = (B + 255) // 256
    flop = B * 1024 * 2 * 15488 * (1 if a.nograd else 2)
    print("decode%s B=%d: %.3f ms -> %.1f us per round of 256 clouds | %.1f TF/s (%.1f%% of 157.3)"
          % ("" if a.nograd else "+grad", B, ms, ms * 1e3 / rounds, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100))
