#!/usr/bin/env python
"""Train the REFERENCE's ConvolutionalOccupancyNetwork (imported where it lies, make_golden.py's shims) on CPU into a
"trained-like" checkpoint: an occupancy field with a real surface at the configured iso-value.

Why: the reference loads pretrain/convonet.pth (ConvONet/opt_defense.py:64-65, ConvONet/README.md:17), a Google-Drive
download that is not available offline, so every other fixture uses seeded RANDOM weights - a field that never crosses
logit(0.2), in which the optimised points never converge onto a surface (other kNN density, clamp hits, neighbour-list
rebuild rates).  The reference ships no training script for ConvONet; this one minimises the occupancy BCE of
ConvONet/opt_defense.py:213-216's logits against ANALYTIC inside / outside labels of the bench's shape families (sphere,
ellipsoid, box, cylinder, torus, two-box "chair": bench.synth_clouds kinds 0-5), with the encoder input prepared exactly
like the pipeline prepares it (unit-sphere normalisation, then preprocess_pc's centre / bbox scaling x 0.9, a 600-point
subset).  Bounded budget (--minutes); seeds fixed, but CPU training is not bit-reproducible across machines, so the result
is COMMITTED: tests/golden/trained_like_f16.npz holds the state_dict in the checkpoint's own key names, rounded to float16
(the checkpoint IS those rounded values; 3.9 MB).  Build container only (needs /root/reference).

    python tests/golden/train_trained_like.py --minutes 25
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (registers the shims, imports the reference modules)

K = 1024


def shape_sample(rng, kind):
    """(surface points [1024,3] in the shape's own frame, inside(q) -> bool) for bench.synth_clouds kind 0-5."""
    def unit(v):
        return v / np.linalg.norm(v, axis=1, keepdims=True)

    def box_surface(ext, m):
        ext = np.asarray(ext, np.float64)
        area = np.array([ext[1] * ext[2], ext[0] * ext[2], ext[0] * ext[1]])
        ax = rng.choice(3, m, p=area / area.sum())
        p = rng.uniform(-1, 1, (m, 3)) * ext
        p[np.arange(m), ax] = rng.choice([-1.0, 1.0], m) * ext[ax]
        return p

    if kind == 0:
        return unit(rng.standard_normal((K, 3))), lambda q: (q ** 2).sum(1) < 1.0
    if kind == 1:
        ax = rng.uniform(0.4, 1.0, 3)
        return unit(rng.standard_normal((K, 3))) * ax, lambda q: ((q / ax) ** 2).sum(1) < 1.0
    if kind == 2:
        ext = rng.uniform(0.3, 1.0, 3)
        return box_surface(ext, K), lambda q: (np.abs(q) < ext).all(1)
    if kind == 3:
        a, h = rng.uniform(0, 2 * np.pi, K), rng.uniform(-1, 1, K)
        r = rng.uniform(0.3, 0.8)
        return np.stack([r * np.cos(a), r * np.sin(a), h], 1), lambda q: (q[:, 0] ** 2 + q[:, 1] ** 2 < r * r) & (np.abs(q[:, 2]) < 1.0)
    if kind == 4:
        a, b = rng.uniform(0, 2 * np.pi, (2, K))
        t = rng.uniform(0.2, 0.4)
        p = np.stack([(1 + t * np.cos(b)) * np.cos(a), (1 + t * np.cos(b)) * np.sin(a), t * np.sin(b)], 1)
        return p, lambda q: (np.sqrt(q[:, 0] ** 2 + q[:, 1] ** 2) - 1.0) ** 2 + q[:, 2] ** 2 < t * t
    m = K // 2
    e1, e2, off = np.array([0.5, 0.5, 0.08]), np.array([0.5, 0.08, 0.5]), np.array([0, 0.45, 0.45])
    p = np.concatenate([box_surface(e1, m), box_surface(e2, K - m) + off])
    return p, lambda q: (np.abs(q) < e1).all(1) | (np.abs(q - off) < e2).all(1)


def make_batch(rng, B, n_query):
    sel, qs, lab = [], [], []
    for _ in range(B):
        p, inside = shape_sample(rng, int(rng.integers(0, 6)))
        m = p.mean(0)
        p1 = p - m
        maxnorm = np.linalg.norm(p1, axis=1).max()
        p1 = p1 / maxnorm                                            # ModelNet40-style unit sphere (bench.synth_clouds)
        ext = (p1.max(0) - p1.min(0)).max()
        c1 = p1.mean(0)                                               # preprocess_pc's centre (0 up to rounding) and bbox scale
        p2 = (p1 - c1) / ext * 0.9
        to_shape = lambda q: (q * ext / 0.9 + c1) * maxnorm + m       # pipeline frame -> the shape's own frame
        pick = rng.choice(K, 600, replace=False)
        sel.append((p2[pick] + rng.normal(0, 0.004, (600, 3))).astype(np.float32))
        qu = rng.uniform(-0.55, 0.55, (n_query // 2, 3))
        qn = p2[rng.integers(0, K, n_query - n_query // 2)] + rng.normal(0, 0.04, (n_query - n_query // 2, 3))
        q = np.concatenate([qu, qn])
        qs.append(q.astype(np.float32))
        lab.append(inside(to_shape(q)).astype(np.float32))
    return torch.from_numpy(np.stack(sel)), torch.from_numpy(np.stack(qs)), torch.from_numpy(np.stack(lab))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=25.0)
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--queries", type=int, default=2048)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--out", default=os.path.join(HERE, "trained_like_f16.npz"))
    a = ap.parse_args()
    torch.manual_seed(0)
    rng = np.random.default_rng(2026)
    enc = MG.LocalPoolPointnet(c_dim=32, dim=3, hidden_dim=32, scatter_type="max", unet=True,
                               unet_kwargs=dict(depth=4, merge_mode="concat", start_filts=32),
                               plane_resolution=64, plane_type=["xz", "xy", "yz"], padding=0.1)
    dec = MG.LocalDecoder(dim=3, c_dim=32, hidden_size=32, sample_mode="bilinear", padding=0.1)
    model = MG.ConvolutionalOccupancyNetwork(dec, enc, device=torch.device("cpu"))
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=a.lr)
    t0, it = time.time(), 0
    ema_loss = ema_acc = None
    while time.time() - t0 < a.minutes * 60:
        sel, q, lab = make_batch(rng, a.batch, a.queries)
        c = model.encode_inputs(sel)
        logits = model.decode(q, c).logits
        loss = F.binary_cross_entropy_with_logits(logits, lab)
        opt.zero_grad()
        loss.backward()
        opt.step()
        acc = float(((logits > 0) == (lab > 0.5)).float().mean())
        ema_loss = float(loss) if ema_loss is None else 0.98 * ema_loss + 0.02 * float(loss)
        ema_acc = acc if ema_acc is None else 0.98 * ema_acc + 0.02 * acc
        it += 1
        if it % 25 == 0:
            print("it %5d  %.1f min  loss %.4f  acc %.4f" % (it, (time.time() - t0) / 60, ema_loss, ema_acc), flush=True)
        if it % 200 == 0:
            save(model, a.out)
    save(model, a.out)
    print("done: %d iterations, loss %.4f, accuracy %.4f" % (it, ema_loss, ema_acc))


def save(model, path):
    sd = {k: v.detach().numpy().astype(np.float16) for k, v in model.state_dict().items()}
    np.savez(path, **sd)


if __name__ == "__main__":
    main()
