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


def pack_state_dict(state: Dict[str, object]) -> np.ndarray:
    """Flatten a state_dict (torch tensors or numpy arrays) into one float32 vector.  Missing or
    mis-shaped tensors raise KeyError / ValueError, like ``load_state_dict(strict=True)``."""
    parts = []
    for name, shape in canonical_keys():
        if name not in state:
            raise KeyError("checkpoint lacks %r" % name)
        t = state[name]
        a = t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
        if tuple(a.shape) != tuple(shape):
            raise ValueError("%s: expected shape %s, got %s" % (name, shape, tuple(a.shape)))
        parts.append(np.ascontiguousarray(a, dtype=np.float32).reshape(-1))
    return np.concatenate(parts)


def load_checkpoint(path: str) -> np.ndarray:
    """``torch.load`` a reference checkpoint (ConvONet/opt_defense.py:65) and pack it.  Accepts a bare
    state_dict or the training checkpoints' {'model': state_dict, ...} wrapper."""
    import torch
    sd = torch.load(path, map_location="cpu")
    if isinstance(sd, dict) and "model" in sd and "decoder.fc_p.weight" not in sd:
        sd = sd["model"]
    return pack_state_dict(sd)
