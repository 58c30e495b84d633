"""Checkpoint handling: the reference's ``state_dict`` -> the canonical flat order of include/ifd.h."""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np


def canonical_keys() -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) of every tensor, in the order ifd_create expects (names as in pretrain/convonet.pth;
    schema: SURVEY.md section 8 R0, ConvONet/src/conv_onet/models/decoder.py:29-40,
    src/encoder/pointnet.py:37-47, src/encoder/unet.py:184-209)."""
    k: List[Tuple[str, Tuple[int, ...]]] = []
    k += [("decoder.fc_p.weight", (32, 3)), ("decoder.fc_p.bias", (32,))]
    for i in range(5):
        k += [(f"decoder.fc_c.{i}.weight", (32, 32)), (f"decoder.fc_c.{i}.bias", (32,))]
    for i in range(5):
        k += [(f"decoder.blocks.{i}.fc_0.weight", (32, 32)), (f"decoder.blocks.{i}.fc_0.bias", (32,)),
              (f"decoder.blocks.{i}.fc_1.weight", (32, 32)), (f"decoder.blocks.{i}.fc_1.bias", (32,))]
    k += [("decoder.fc_out.weight", (1, 32)), ("decoder.fc_out.bias", (1,))]
    k += [("encoder.fc_pos.weight", (64, 3)), ("encoder.fc_pos.bias", (64,))]
    for i in range(5):
        k += [(f"encoder.blocks.{i}.fc_0.weight", (32, 64)), (f"encoder.blocks.{i}.fc_0.bias", (32,)),
              (f"encoder.blocks.{i}.fc_1.weight", (32, 32)), (f"encoder.blocks.{i}.fc_1.bias", (32,)),
              (f"encoder.blocks.{i}.shortcut.weight", (32, 64))]
    k += [("encoder.fc_c.weight", (32, 32)), ("encoder.fc_c.bias", (32,))]
    chans = [32, 64, 128, 256]
    cin = 32
    for i, co in enumerate(chans):
        k += [(f"encoder.unet.down_convs.{i}.conv1.weight", (co, cin, 3, 3)),
              (f"encoder.unet.down_convs.{i}.conv1.bias", (co,)),
              (f"encoder.unet.down_convs.{i}.conv2.weight", (co, co, 3, 3)),
              (f"encoder.unet.down_convs.{i}.conv2.bias", (co,))]
        cin = co
    for i in range(3):
        co = cin // 2
        k += [(f"encoder.unet.up_convs.{i}.upconv.weight", (cin, co, 2, 2)),
              (f"encoder.unet.up_convs.{i}.upconv.bias", (co,)),
              (f"encoder.unet.up_convs.{i}.conv1.weight", (co, 2 * co, 3, 3)),
              (f"encoder.unet.up_convs.{i}.conv1.bias", (co,)),
              (f"encoder.unet.up_convs.{i}.conv2.weight", (co, co, 3, 3)),
              (f"encoder.unet.up_convs.{i}.conv2.bias", (co,))]
        cin = co
    k += [("encoder.unet.conv_final.weight", (32, 32, 1, 1)), ("encoder.unet.conv_final.bias", (32,))]
    return k


def onet_canonical_keys() -> List[Tuple[str, Tuple[int, ...]]]:
    """ONet-Opt (ONet/configs/onet_mn40.yaml): the reference checkpoint's state_dict order without the BatchNorm
    ``num_batches_tracked`` scalars (include/ifd.h, ONet section).  Conv1d kernels keep their trailing 1."""
    k: List[Tuple[str, Tuple[int, ...]]] = [("decoder.fc_p.weight", (256, 3, 1)), ("decoder.fc_p.bias", (256,))]

    def cbn(pre):
        return [(pre + ".conv_gamma.weight", (256, 512, 1)), (pre + ".conv_gamma.bias", (256,)),
                (pre + ".conv_beta.weight", (256, 512, 1)), (pre + ".conv_beta.bias", (256,)),
                (pre + ".bn.running_mean", (256,)), (pre + ".bn.running_var", (256,))]

    for i in range(5):
        k += cbn(f"decoder.block{i}.bn_0") + cbn(f"decoder.block{i}.bn_1")
        k += [(f"decoder.block{i}.fc_0.weight", (256, 256, 1)), (f"decoder.block{i}.fc_0.bias", (256,)),
              (f"decoder.block{i}.fc_1.weight", (256, 256, 1)), (f"decoder.block{i}.fc_1.bias", (256,))]
    k += cbn("decoder.bn") + [("decoder.fc_out.weight", (1, 256, 1)), ("decoder.fc_out.bias", (1,))]
    k += [("encoder.fc_pos.weight", (1024, 3)), ("encoder.fc_pos.bias", (1024,))]
    for i in range(5):
        k += [(f"encoder.block_{i}.fc_0.weight", (512, 1024)), (f"encoder.block_{i}.fc_0.bias", (512,)),
              (f"encoder.block_{i}.fc_1.weight", (512, 512)), (f"encoder.block_{i}.fc_1.bias", (512,)),
              (f"encoder.block_{i}.shortcut.weight", (512, 1024))]
    k += [("encoder.fc_c.weight", (512, 512)), ("encoder.fc_c.bias", (512,))]
    return k


def pack_state_dict(state: Dict[str, object], model: str = "convonet") -> np.ndarray:
    """Flatten a state_dict (torch tensors or numpy arrays) into one float32 vector.  Missing or
    mis-shaped tensors raise KeyError / ValueError, like ``load_state_dict(strict=True)``."""
    parts = []
    for name, shape in (onet_canonical_keys() if model == "onet" else canonical_keys()):
        if name not in state:
            raise KeyError("checkpoint lacks %r" % name)
        t = state[name]
        a = t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
        if tuple(a.shape) != tuple(shape):
            raise ValueError("%s: expected shape %s, got %s" % (name, shape, tuple(a.shape)))
        parts.append(np.ascontiguousarray(a, dtype=np.float32).reshape(-1))
    return np.concatenate(parts)


def load_checkpoint(path: str, model: str = "convonet") -> np.ndarray:
    """``torch.load`` a reference checkpoint (ConvONet/opt_defense.py:65) and pack it.  Accepts a bare
    state_dict or the training checkpoints' {'model': state_dict, ...} wrapper."""
    import torch
    sd = torch.load(path, map_location="cpu")
    if isinstance(sd, dict) and "model" in sd and "decoder.fc_p.weight" not in sd:
        sd = sd["model"]
    return pack_state_dict(sd, model)


def random_state_dict(seed: int = 0) -> Dict[str, np.ndarray]:
    """Seeded random weights of the shipped architecture (benchmarks / smoke runs; the trained
    pretrain/convonet.pth is a Google-Drive download).  Linear layers: U(-1/sqrt(fan_in), +); convs:
    Xavier-normal with a small uniform bias; fc_1 is NOT zero-initialised (src/layers.py:37 would make
    every residual branch constant)."""
    import math
    rng = np.random.default_rng(seed)
    w: Dict[str, np.ndarray] = {}
    def uni(shape, bound):
        return rng.uniform(-bound, bound, size=shape).astype(np.float32)

    def linear(name, n_out, n_in, bias=True):
        b = 1.0 / math.sqrt(n_in)
        w[name + ".weight"] = uni((n_out, n_in), b)
        if bias:
            w[name + ".bias"] = uni((n_out,), b)

    def conv(name, c_out, c_in, k, transpose=False):
        shape = (c_in, c_out, k, k) if transpose else (c_out, c_in, k, k)
        std = math.sqrt(2.0 / (c_in * k * k + c_out * k * k))
        w[name + ".weight"] = (rng.standard_normal(shape) * std).astype(np.float32)
        w[name + ".bias"] = uni((c_out,), 0.05)

    linear("decoder.fc_p", 32, 3)
    for i in range(5):
        linear(f"decoder.fc_c.{i}", 32, 32)
    for i in range(5):
        linear(f"decoder.blocks.{i}.fc_0", 32, 32)
        linear(f"decoder.blocks.{i}.fc_1", 32, 32)
    linear("decoder.fc_out", 1, 32)
    linear("encoder.fc_pos", 64, 3)
    for i in range(5):
        linear(f"encoder.blocks.{i}.fc_0", 32, 64)
        linear(f"encoder.blocks.{i}.fc_1", 32, 32)
        linear(f"encoder.blocks.{i}.shortcut", 32, 64, bias=False)
    linear("encoder.fc_c", 32, 32)
    c_in = 32
    for i, c_out in enumerate([32, 64, 128, 256]):
        conv(f"encoder.unet.down_convs.{i}.conv1", c_out, c_in, 3)
        conv(f"encoder.unet.down_convs.{i}.conv2", c_out, c_out, 3)
        c_in = c_out
    for i in range(3):
        c_out = c_in // 2
        conv(f"encoder.unet.up_convs.{i}.upconv", c_out, c_in, 2, transpose=True)
        conv(f"encoder.unet.up_convs.{i}.conv1", c_out, 2 * c_out, 3)
        conv(f"encoder.unet.up_convs.{i}.conv2", c_out, c_out, 3)
        c_in = c_out
    conv("encoder.unet.conv_final", 32, c_in, 1)
    return w


def onet_random_state_dict(seed: int = 0) -> Dict[str, np.ndarray]:
    """Seeded random ONet weights (same streams as the test oracle's: Linear / 1x1-conv U(+-1/sqrt(fan_in)); what
    the reference zero-initialises - fc_1, CBN gamma/beta convs - is randomised; non-trivial BatchNorm statistics)."""
    rng = np.random.default_rng(seed + 7001)
    w: Dict[str, np.ndarray] = {}

    def lin(key, n_out, n_in, bias=True, conv=False, scale=1.0):
        b = scale / np.sqrt(n_in)
        shape = (n_out, n_in, 1) if conv else (n_out, n_in)
        w[key + ".weight"] = rng.uniform(-b, b, size=shape).astype(np.float32)
        if bias:
            w[key + ".bias"] = rng.uniform(-b, b, size=(n_out,)).astype(np.float32)

    lin("encoder.fc_pos", 1024, 3)
    for i in range(5):
        lin(f"encoder.block_{i}.fc_0", 512, 1024)
        lin(f"encoder.block_{i}.fc_1", 512, 512)
        lin(f"encoder.block_{i}.shortcut", 512, 1024, bias=False)
    lin("encoder.fc_c", 512, 512)

    def cbn(key):
        lin(key + ".conv_gamma", 256, 512, conv=True, scale=8.0)
        w[key + ".conv_gamma.bias"] = (1.0 + rng.uniform(-0.2, 0.2, 256)).astype(np.float32)
        lin(key + ".conv_beta", 256, 512, conv=True, scale=8.0)
        w[key + ".bn.running_mean"] = rng.normal(0.0, 0.3, 256).astype(np.float32)
        w[key + ".bn.running_var"] = rng.uniform(0.5, 1.5, 256).astype(np.float32)

    lin("decoder.fc_p", 256, 3, conv=True)
    for i in range(5):
        cbn(f"decoder.block{i}.bn_0")
        cbn(f"decoder.block{i}.bn_1")
        lin(f"decoder.block{i}.fc_0", 256, 256, conv=True)
        lin(f"decoder.block{i}.fc_1", 256, 256, conv=True)
    cbn("decoder.bn")
    lin("decoder.fc_out", 1, 256, conv=True)
    return w
