"""Round-5 verdict, item 2: does point ORDER (locality of a tile's taps) move the optimiser launch?  Same clouds, same planes, the
initial points of every cloud in three orders:
  A  as drawn            (torch.randint draws: a tile's 32 points are scattered over the surface)
  B  Morton order        (30-bit Z-curve of the coordinates: a tile's 32 points are neighbours on the surface, the 12 x 32 taps of a
                          tile fall into a few rows of each plane)
  C  plane-cell order    (sorted by the xz cell, then y: taps of plane 0 of a tile share rows AND lines where points share a cell)
The optimiser is index-agnostic (exact 5-NN sets; ties by index are the only order dependence), so B / C compute the same
restoration on a permuted cloud; what changes is which lines a wave's gathers touch together.  f32 and bf16x6 / bf16x3, best of 3.
Prints launch times and (ifd_get_counters) list statistics; L2 hit rates come from scripts/pmc_locality.sh over the same script.
    python scripts/ab_locality.py [clouds] [modes]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import ifdefense_amd as I

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2468
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["f32", "bf16x6", "bf16x3"]
orders = os.environ.get("IFD_LOCALITY_ORDERS", "A,B,C").split(",")
r = I.Restorer(I.weights.pack_state_dict(I.weights.random_state_dict(0)), device="cuda:0")
x = torch.from_numpy(bench.synth_clouds(n)).cuda()
prep = r.prepare(x, r.sor(x), seed=1234)
planes = r.encode_inputs(prep["sel"], prep["t_per_cloud"])
lb = torch.full((n,), 192, dtype=torch.int32, device="cuda")
init = prep["init"]


def spread3(v):
    v = v.to(torch.int64) & 0x3FF
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v


def morton_perm(p):
    q = ((p + 0.6) / 1.2 * 1023.0).clamp(0, 1023).to(torch.int64)
    code = spread3(q[..., 0]) | (spread3(q[..., 1]) << 1) | (spread3(q[..., 2]) << 2)
    return code.argsort(dim=1)


def cell_perm(p):
    u = ((p / 1.10001 + 0.5).clamp(0, 1 - 1e-5) * 63.0).floor().to(torch.int64)        # grid_sample pixel cell (align_corners)
    key = (u[..., 2] * 64 + u[..., 0]) * 64 + u[..., 1]                                   # xz cell (row = z), then y
    return key.argsort(dim=1)


def permuted(perm):
    return torch.gather(init, 1, perm[..., None].expand(-1, -1, 3)).contiguous()


inits = {"A": init}
if "B" in orders:
    inits["B"] = permuted(morton_perm(init))
if "C" in orders:
    inits["C"] = permuted(cell_perm(init))


def run(p0, mode):
    best, out = None, None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = r.optimize_points(p0, planes, rep_weight=500.0, iterations=500, loss_batch=lb, split=1, precision=mode)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    return best, out


for mode in modes:
    res = {}
    for name in orders:
        ms, out = run(inits[name], mode)
        c = r.counters()
        res[name] = ms
        print("%-7s order %s: %8.1f ms  %7.1f clouds/s   list rebuilds per cloud %.1f, ring on %.2f of the wave-steps, finite %s" %
              (mode, name, ms, n / ms * 1e3, c["knn_rebuilds"] / 8.0 / n, c["knn_ring_evals"] / (8.0 * n * 501), bool(torch.isfinite(out).all())))
    base = res[orders[0]]
    print("%-7s relative to order %s: %s" % (mode, orders[0], ", ".join("%s %+.2f %%" % (k, 100.0 * (v - base) / base) for k, v in res.items())))
