"""Reference-style GPU baseline: the oracle's op sequence - bmm kNN + topk, autograd through
grid_sample and the decoder MLP, torch.optim.Adam, one small kernel per op - executed by PyTorch-ROCm on one MI355X.
This is what running the reference's own Python on the GPU amounts to; it is a reported baseline, not a test
(pytest does not collect it) and not part of the product.  Usage: python tests/perf_reference_style_gpu.py [clouds] [steps]
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from oracle import convonet_oracle as O  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 192          # the reference's batch size
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    dev = torch.device("cuda:0")
    w = {k: v.to(dev) for k, v in O.to_torch(O.make_random_weights(0)).items()}
    clouds = bench.synth_clouds(n)
    t0 = time.perf_counter()
    keep, _ = O.sor_keep_mask(torch.from_numpy(clouds).to(dev))
    keep = keep.cpu().numpy().astype(bool)
    procs = [O.preprocess_pc(clouds[b][keep[b]]) for b in range(n)]
    g = torch.Generator().manual_seed(0)
    sel = torch.stack([torch.from_numpy(p[torch.randperm(len(p), generator=g)[:600].numpy()]) for p in procs]).to(dev)
    init = torch.stack([torch.from_numpy(p[torch.randint(len(p), (1024,), generator=g).numpy()]) for p in procs])
    init = (init + 0.01 * torch.randn(init.shape, generator=g)).clamp(-0.45, 0.45).to(dev)
    with torch.no_grad():
        planes = O.encode_inputs(w, sel)
    torch.cuda.synchronize()
    t_pre = time.perf_counter() - t0
    O.optimize_points(w, init[:2], {k: v[:2] for k, v in planes.items()}, iterations=1)      # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    O.optimize_points(w, init, planes, rep_weight=500.0, iterations=steps - 1)
    torch.cuda.synchronize()
    t_opt = time.perf_counter() - t0
    total = t_pre + t_opt * 501.0 / steps
    print(json.dumps({"baseline": "unfused PyTorch-ROCm (reference-style op sequence)", "clouds": n, "steps_timed": steps,
                      "ms_per_step": 1e3 * t_opt / steps, "pre_s": t_pre, "clouds_per_s_scaled_to_501": n / total}))


if __name__ == "__main__":
    main()
