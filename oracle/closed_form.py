"""float64 closed-form evaluation of one ConvONet-Opt objective gradient.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  No autograd: the forward
and the hand-derived backward of SURVEY.md Appendix A, written with numpy in
float64.  It is the "ground truth" the float32 implementations (the torch
oracle and the HIP kernels) are both compared with, and it exposes the
intermediate quantities (logits, kNN sets, the two gradient parts).

Reference lines restated: decoder.py:50-95 (sampling + MLP), layers.py:39-48,
common.py:235-258, opt_defense.py:212-225 (BCE to threshold, 1/B scaling),
repulsion_loss.py:43-54, pn_utils.py:64-83.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

PADDING = 0.1
RES = 64
PLANES = ("xz", "xy", "yz")
PLANE_AXES = {"xz": (0, 2), "xy": (0, 1), "yz": (1, 2)}
S_DIV = 1 + PADDING + 10e-6


def _plane_uv(x: np.ndarray, plane: str):
    a0, a1 = PLANE_AXES[plane]
    u = np.stack((x[:, a0], x[:, a1]), axis=-1) / S_DIV + 0.5
    live = np.ones_like(u)
    hi, lo = u >= 1, u < 0
    u = np.where(hi, 1 - 10e-6, u)
    u = np.where(lo, 0.0, u)
    live[hi | lo] = 0.0
    return u, live, (a0, a1)


def _bilinear(plane_chw: np.ndarray, u: np.ndarray):
    """plane [C,H,W]; u [K,2] -> c [K,C], dc/du0 [K,C], dc/du1 [K,C]."""
    ix = np.clip(u[:, 0] * (RES - 1), 0, RES - 1)
    iy = np.clip(u[:, 1] * (RES - 1), 0, RES - 1)
    x0 = np.minimum(np.floor(ix).astype(np.int64), RES - 2)
    y0 = np.minimum(np.floor(iy).astype(np.int64), RES - 2)
    tx, ty = (ix - x0)[:, None], (iy - y0)[:, None]
    nw = plane_chw[:, y0, x0].T
    ne = plane_chw[:, y0, x0 + 1].T
    sw = plane_chw[:, y0 + 1, x0].T
    se = plane_chw[:, y0 + 1, x0 + 1].T
    c = nw * (1 - tx) * (1 - ty) + ne * tx * (1 - ty) + sw * (1 - tx) * ty + se * tx * ty
    dcdu0 = (RES - 1) * ((ne - nw) * (1 - ty) + (se - sw) * ty)
    dcdu1 = (RES - 1) * ((sw - nw) * (1 - tx) + (se - ne) * tx)
    return c, dcdu0, dcdu1


def decoder_forward_backward(w: Dict[str, np.ndarray], x: np.ndarray, planes: Dict[str, np.ndarray],
                             threshold: float = 0.2, loss_batch: float = 1.0):
    """One cloud.  x [K,3]; planes[pl] [C,H,W].  Returns dict(logits, occ_loss_sum, grad [K,3])."""
    W = {k: np.asarray(v, np.float64) for k, v in w.items() if k.startswith("decoder.")}
    x = np.asarray(x, np.float64)
    samples = {}
    c = 0.0
    for pl in PLANES:
        u, live, axes = _plane_uv(x, pl)
        cp, d0, d1 = _bilinear(np.asarray(planes[pl], np.float64), u)
        samples[pl] = (d0, d1, live, axes)
        c = c + cp
    n = x @ W["decoder.fc_p.weight"].T + W["decoder.fc_p.bias"]
    acts = []
    for i in range(5):
        a = n + c @ W[f"decoder.fc_c.{i}.weight"].T + W[f"decoder.fc_c.{i}.bias"]
        h = np.maximum(a, 0) @ W[f"decoder.blocks.{i}.fc_0.weight"].T + W[f"decoder.blocks.{i}.fc_0.bias"]
        n = a + np.maximum(h, 0) @ W[f"decoder.blocks.{i}.fc_1.weight"].T + W[f"decoder.blocks.{i}.fc_1.bias"]
        acts.append((a, h))
    logit = np.maximum(n, 0) @ W["decoder.fc_out.weight"][0] + W["decoder.fc_out.bias"][0]
    bce = np.maximum(logit, 0) - threshold * logit + np.log1p(np.exp(-np.abs(logit)))
    dlogit = (1.0 / (1.0 + np.exp(-logit)) - threshold) / loss_batch

    dn = dlogit[:, None] * W["decoder.fc_out.weight"][0][None, :] * (n > 0)
    dc = np.zeros_like(dn)
    for i in range(4, -1, -1):
        a, h = acts[i]
        dh = (dn @ W[f"decoder.blocks.{i}.fc_1.weight"]) * (h > 0)
        da = dn + (dh @ W[f"decoder.blocks.{i}.fc_0.weight"]) * (a > 0)
        dc = dc + da @ W[f"decoder.fc_c.{i}.weight"]
        dn = da
    grad = dn @ W["decoder.fc_p.weight"]
    for pl in PLANES:
        d0, d1, live, (a0, a1) = samples[pl]
        grad[:, a0] += np.sum(dc * d0, axis=1) * live[:, 0] / S_DIV
        grad[:, a1] += np.sum(dc * d1, axis=1) * live[:, 1] / S_DIV
    return {"logits": logit, "occ_loss_sum": bce.sum(), "grad": grad}


def knn_exact(x: np.ndarray, k: int = 5) -> np.ndarray:
    """Exact direct-form kNN in float64 (self excluded by index): [K,3] -> [K,k] sorted by distance."""
    x = np.asarray(x, np.float64)
    d2 = ((x[:, None, :] - x[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d2, np.inf)
    return np.argsort(d2, axis=1, kind="stable")[:, :k]


def repulsion_forward_backward(x: np.ndarray, idx: Optional[np.ndarray] = None, rep_weight: float = 500.0,
                               loss_batch: float = 1.0, radius: float = 0.07, h: float = 0.03,
                               eps: float = 1e-12):
    """One cloud.  Returns dict(rep_mean (un-weighted mean over K*5), grad [K,3] incl. rep_weight/loss_batch)."""
    x = np.asarray(x, np.float64)
    K = x.shape[0]
    if idx is None:
        idx = knn_exact(x)
    k = idx.shape[1]
    diff = x[idx] - x[:, None, :]                       # [K,k,3]
    d2raw = (diff ** 2).sum(-1)
    d2 = np.maximum(d2raw, eps)
    d = np.sqrt(d2)
    wgt = np.exp(-d2 / (h * h))
    rep_mean = ((radius - d) * wgt).mean()
    scale = rep_weight / (loss_batch * K * k)
    dL_dd = scale * (-wgt - (radius - d) * wgt * 2 * d / (h * h))
    coef = np.where(d2raw > eps, dL_dd / d, 0.0)        # clamp(min=eps) kills the gradient below eps
    g = coef[..., None] * diff                          # dL/dx_j contribution; centre gets -g
    grad = -g.sum(axis=1)
    np.add.at(grad, idx.reshape(-1), g.reshape(-1, 3))
    return {"rep_mean": rep_mean, "grad": grad, "idx": idx}
