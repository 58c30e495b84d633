#!/usr/bin/env python
"""The reference's OWN driver functions, executed from its source file - not restated.

``ConvONet/opt_defense.py`` cannot be imported (argparse, ``torch.device("cuda")`` and ``torch.load`` of an absent
checkpoint run at import time, :21-73).  So the file is parsed and only the ``FunctionDef`` nodes of

    normalize_batch_pc (:76-83)   sor_process (:86-111)   preprocess_pc (:114-146)   init_points (:149-179)
    optimize_points (:182-239)    defend_point_cloud (:255-314)

are compiled and executed - the reference's statements, byte for byte - in a namespace that supplies what the module
level would have supplied: ``args`` (the CLI namespace), ``generator`` (holder of ``.model``), ``repulsion_loss``,
``SORDefense``, ``tqdm``, ``F``, and ``np`` / ``torch`` proxies that forward everything except
  * the three unseeded random draws (``np.random.choice``, ``torch.randint``, ``torch.randn_like``), which return RECORDED
    draws when a ``Draws`` object carries them (and record what they drew otherwise) - the fixtures are RNG-independent;
  * ``torch.optim.Adam``, replaced by a subclass that snapshots (x, grad, exp_avg, exp_avg_sq, x_next) at chosen steps -
    the arithmetic is torch.optim.Adam's own ``step``.
``Tensor.cuda`` is a no-op (make_golden.py's shim).  Build container only: reads /root/reference.
"""
import ast
import types

import numpy as np
import torch
import torch.nn.functional as F

REF_FILE = "/root/reference/ConvONet/opt_defense.py"
WANT = ("normalize_batch_pc", "sor_process", "preprocess_pc", "init_points", "optimize_points", "defend_point_cloud")


class Draws:
    """Recorded random draws, consumed in call order (or filled in when ``record``)."""

    def __init__(self, choice=None, randint=None, randn=None):
        self.choice = list(choice) if choice is not None else None       # np.random.choice results (600-subsets)
        self.randint = list(randint) if randint is not None else None    # torch.randint results (init indices)
        self.randn = list(randn) if randn is not None else None          # torch.randn_like results (init noise)
        self.log = {"choice": [], "randint": [], "randn": []}

    def take(self, kind, fresh):
        src = getattr(self, kind)
        v = fresh() if src is None else src.pop(0)
        self.log[kind].append(v)
        return v


class RecordingAdam(torch.optim.Adam):
    """torch.optim.Adam with snapshots; ``record`` = step indices (0-based, the reference's loop variable i)."""
    record = ()
    snaps = None
    last_losses = None

    def step(self, closure=None):
        cls = type(self)
        i = getattr(self, "_i", 0)
        (x,) = self.param_groups[0]["params"]
        if i in cls.record:
            st = self.state[x]
            cls.snaps[i] = dict(x=x.detach().clone(), g=x.grad.detach().clone(),
                                m=st["exp_avg"].clone() if st else torch.zeros_like(x),
                                v=st["exp_avg_sq"].clone() if st else torch.zeros_like(x))
            if cls.last_losses is not None:
                cls.snaps[i].update(cls.last_losses)
        out = super().step(closure)
        if i in cls.record:
            cls.snaps[i]["x_next"] = x.detach().clone()
        cls.snaps["final_unnormalised"] = x.detach().clone()
        self._i = i + 1
        return out


class _Proxy:
    def __init__(self, target, **override):
        self.__dict__["_t"] = target
        self.__dict__.update(override)

    def __getattr__(self, name):
        return getattr(self._t, name)


def load(model, args, draws=None, rep_loss_fn=None):
    """Namespace with the reference's four functions bound to `model` / `args`.  ns["draws"] is the Draws in use."""
    import tqdm
    from defense import SORDefense
    from defense import repulsion_loss as ref_rep
    draws = draws or Draws()
    tree = ast.parse(open(REF_FILE).read(), REF_FILE)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANT]
    assert sorted(n.name for n in body) == sorted(WANT), [n.name for n in body]
    mod = ast.Module(body=body, type_ignores=[])

    def choice(n, size, replace=True):
        return draws.take("choice", lambda: np.random.choice(n, size, replace=replace))

    def randint(lo, hi, shape):
        return draws.take("randint", lambda: torch.randint(lo, hi, shape))

    def randn_like(t):
        return draws.take("randn", lambda: torch.randn_like(t))

    losses = {}

    def bce(logits, target, reduction="mean"):
        out = F.binary_cross_entropy_with_logits(logits, target, reduction=reduction)
        losses["occ"] = float(out.detach().mean() * logits.shape[1])           # what :215-216 make of it
        RecordingAdam.last_losses = losses
        return out

    rep_inner = rep_loss_fn or ref_rep

    def rep(points):
        out = rep_inner(points)
        losses["rep_mean"] = float(out.detach().mean())                        # x rep_weight at :223
        return out

    np_proxy = _Proxy(np, random=_Proxy(np.random, choice=choice))
    torch_proxy = _Proxy(torch, randint=randint, randn_like=randn_like, optim=_Proxy(torch.optim, Adam=RecordingAdam))
    ns = {"np": np_proxy, "torch": torch_proxy, "F": _Proxy(F, binary_cross_entropy_with_logits=bce), "args": args,
          "generator": types.SimpleNamespace(model=model), "repulsion_loss": rep, "SORDefense": SORDefense, "tqdm": tqdm,
          "draws": draws}
    exec(compile(mod, REF_FILE, "exec"), ns)
    return ns


def run_optimize(ns, init, planes, iterations, rep_weight=500.0, record=()):
    """optimize_points(opt_points, None, c, rep_weight, iterations) of the reference with snapshots at `record`.
    Returns (the function's own return value: normalised numpy [B,K,3], snapshots incl. "final_unnormalised")."""
    RecordingAdam.record, RecordingAdam.snaps, RecordingAdam.last_losses = tuple(record), {}, None
    out = ns["optimize_points"](init.clone(), None, planes, rep_weight=rep_weight, iterations=iterations)
    snaps = RecordingAdam.snaps
    for i in record:
        snaps[i]["rep"] = snaps[i].pop("rep_mean") * rep_weight
    return out, snaps


def default_args(**kw):
    a = dict(sample_npoint=1024, padding_scale=0.9, init_sigma=0.01, lr=0.001, threshold=0.2, input_npoint=600,
             iterations=200, batch_size=192, rep_weight=500.0, sor=True, sor_k=2, sor_alpha=1.1)
    a.update(kw)
    return types.SimpleNamespace(**a)
