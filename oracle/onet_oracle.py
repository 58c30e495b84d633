"""PyTorch-CPU float32 restatement of the ONet-Opt restoration path (ONet/opt_defense.py).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  ONet-Opt (BASELINE config #1) is
the same optimisation loop as ConvONet-Opt with a different conditioning: a global 512-d latent code from a
PointNet with ResNet blocks, and a decoder whose five ResNet blocks use conditional batch normalisation (CBN)
in eval mode.  Everything that is identical to ConvONet-Opt (SOR, preprocess, init, repulsion loss, BCE,
Adam, normalisation; ``diff ConvONet/opt_defense.py ONet/opt_defense.py`` touches only the config path, the
``decode(p, z, c)`` call and the save name) is taken from ``convonet_oracle``.

Weights: plain ``dict[str, torch.Tensor]`` with the reference checkpoint's ``state_dict`` key names
(``encoder.*`` / ``decoder.*``, Conv1d weights keep their trailing kernel dimension of 1), so a real
``pretrain/onet.pth`` can be dropped in unchanged.  Every function names the reference lines it restates
(paths relative to /root/reference/ONet).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import convonet_oracle as CO

Weights = Dict[str, torch.Tensor]

# Resolved hyper-parameters of the shipped config (configs/onet_mn40.yaml + configs/default.yaml)
C_DIM = 512              # onet_mn40.yaml:18
ENC_HIDDEN = 512         # onet_mn40.yaml:16-17
DEC_HIDDEN = 256         # onet/models/decoder.py:89 (default hidden_size)
Z_DIM = 0                # onet_mn40.yaml:19  -> no fc_z, z is an empty tensor
N_BLOCKS = 5             # decoder.py:96-100, encoder/pointnet.py:76-80
THRESHOLD = 0.2          # onet_mn40.yaml:34
POINTCLOUD_N = 300       # onet_mn40.yaml:6
BN_EPS = 1e-5            # torch.nn.BatchNorm1d default (layers.py:210)


def make_random_weights(seed: int = 0) -> Dict[str, np.ndarray]:
    """Seeded random weights with the reference architecture and key names.

    Linear / 1x1-conv layers: torch default U(+-1/sqrt(fan_in)).  What the reference zero-initialises
    (``fc_1.weight`` layers.py:37,95; CBN ``conv_gamma/conv_beta.weight`` layers.py:216-217) is randomised
    instead, and the BatchNorm running statistics are non-trivial, so that every term of the trained model's
    arithmetic is exercised.  numpy ``Generator`` streams, like ``convonet_oracle.make_random_weights``.
    """
    rng = np.random.default_rng(seed + 7001)
    w: Dict[str, np.ndarray] = {}

    def lin(key, n_out, n_in, bias=True, conv=False, scale=1.0):
        b = scale / np.sqrt(n_in)
        shape = (n_out, n_in, 1) if conv else (n_out, n_in)
        w[key + ".weight"] = rng.uniform(-b, b, size=shape).astype(np.float32)
        if bias:
            w[key + ".bias"] = rng.uniform(-b, b, size=(n_out,)).astype(np.float32)

    # encoder: ResnetPointnet(c_dim 512, hidden 512)  (encoder/pointnet.py:60-84)
    lin("encoder.fc_pos", 2 * ENC_HIDDEN, 3)
    for i in range(N_BLOCKS):
        lin(f"encoder.block_{i}.fc_0", ENC_HIDDEN, 2 * ENC_HIDDEN)
        lin(f"encoder.block_{i}.fc_1", ENC_HIDDEN, ENC_HIDDEN)
        lin(f"encoder.block_{i}.shortcut", ENC_HIDDEN, 2 * ENC_HIDDEN, bias=False)
    lin("encoder.fc_c", C_DIM, ENC_HIDDEN)

    # decoder: DecoderCBatchNorm(z_dim 0, c_dim 512, hidden 256)  (onet/models/decoder.py:88-113)
    def cbn(key):
        lin(key + ".conv_gamma", DEC_HIDDEN, C_DIM, conv=True, scale=8.0)
        w[key + ".conv_gamma.bias"] = (1.0 + rng.uniform(-0.2, 0.2, DEC_HIDDEN)).astype(np.float32)
        lin(key + ".conv_beta", DEC_HIDDEN, C_DIM, conv=True, scale=8.0)
        w[key + ".bn.running_mean"] = rng.normal(0.0, 0.3, DEC_HIDDEN).astype(np.float32)
        w[key + ".bn.running_var"] = rng.uniform(0.5, 1.5, DEC_HIDDEN).astype(np.float32)
        w[key + ".bn.num_batches_tracked"] = np.array(1000, np.int64)

    lin("decoder.fc_p", DEC_HIDDEN, 3, conv=True)
    for i in range(N_BLOCKS):
        cbn(f"decoder.block{i}.bn_0")
        cbn(f"decoder.block{i}.bn_1")
        lin(f"decoder.block{i}.fc_0", DEC_HIDDEN, DEC_HIDDEN, conv=True)
        lin(f"decoder.block{i}.fc_1", DEC_HIDDEN, DEC_HIDDEN, conv=True)
    cbn("decoder.bn")
    lin("decoder.fc_out", 1, DEC_HIDDEN, conv=True)
    return w


def to_torch(weights: Dict[str, np.ndarray]) -> Weights:
    return {k: torch.from_numpy(np.asarray(v)) for k, v in weights.items()}


# --------------------------------------------------------------------------
# encoder  (im2mesh/encoder/pointnet.py:60-113, im2mesh/layers.py:6-48)
# --------------------------------------------------------------------------
def _resnet_block_fc(w: Weights, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """ResnetBlockFC (layers.py:39-48): x_s + fc_1(relu(fc_0(relu(x)))), shortcut without bias."""
    net = F.linear(F.relu(x), w[prefix + ".fc_0.weight"], w[prefix + ".fc_0.bias"])
    dx = F.linear(F.relu(net), w[prefix + ".fc_1.weight"], w[prefix + ".fc_1.bias"])
    return F.linear(x, w[prefix + ".shortcut.weight"]) + dx


def encode_latent(w: Weights, p: torch.Tensor, return_stages: bool = False):
    """ResnetPointnet.forward (encoder/pointnet.py:86-113): p [B,T,3] -> c [B,512].

    fc_pos -> block_0 -> 4 x (concat with the max over the cloud's points, block_i) -> max-pool -> fc_c(relu).
    """
    net = F.linear(p, w["encoder.fc_pos.weight"], w["encoder.fc_pos.bias"])
    net = _resnet_block_fc(w, "encoder.block_0", net)
    stages = [net]
    for i in range(1, N_BLOCKS):
        pooled = net.max(dim=1, keepdim=True).values.expand(net.size())
        net = _resnet_block_fc(w, f"encoder.block_{i}", torch.cat([net, pooled], dim=2))
        stages.append(net)
    net = net.max(dim=1).values
    c = F.linear(F.relu(net), w["encoder.fc_c.weight"], w["encoder.fc_c.bias"])
    return (c, stages) if return_stages else c


# --------------------------------------------------------------------------
# decoder  (im2mesh/onet/models/decoder.py:77-133, im2mesh/layers.py:51-107,193-242)
# --------------------------------------------------------------------------
def cbn_affine(w: Weights, prefix: str, c: torch.Tensor):
    """CBatchNorm1d in eval mode (layers.py:221-242) as a per-cloud, per-channel affine map.

    out = gamma(c) * (x - running_mean) / sqrt(running_var + eps) + beta(c)  ==  a * x + b
    with a = gamma / sqrt(var + eps), b = beta - a * mean; gamma / beta are 1x1 convs of c (layers.py:234-235).
    """
    gamma = F.linear(c, w[prefix + ".conv_gamma.weight"].squeeze(-1), w[prefix + ".conv_gamma.bias"])
    beta = F.linear(c, w[prefix + ".conv_beta.weight"].squeeze(-1), w[prefix + ".conv_beta.bias"])
    inv = 1.0 / torch.sqrt(w[prefix + ".bn.running_var"] + BN_EPS)
    a = gamma * inv
    return a, beta - a * w[prefix + ".bn.running_mean"]


def _cbn(w: Weights, prefix: str, x: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
    """The reference's op order (F.batch_norm in eval mode, then gamma * net + beta); x [B,K,F]."""
    gamma = F.linear(c, w[prefix + ".conv_gamma.weight"].squeeze(-1), w[prefix + ".conv_gamma.bias"])
    beta = F.linear(c, w[prefix + ".conv_beta.weight"].squeeze(-1), w[prefix + ".conv_beta.bias"])
    net = (x - w[prefix + ".bn.running_mean"]) / torch.sqrt(w[prefix + ".bn.running_var"] + BN_EPS)
    return gamma[:, None, :] * net + beta[:, None, :]


def decode_logits(w: Weights, p: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
    """DecoderCBatchNorm.forward with z_dim = 0 (decoder.py:115-133): p [B,K,3], c [B,512] -> logits [B,K]."""
    net = F.linear(p, w["decoder.fc_p.weight"].squeeze(-1), w["decoder.fc_p.bias"])
    for i in range(N_BLOCKS):                                   # CResnetBlockConv1d.forward (layers.py:97-107)
        pre = f"decoder.block{i}"
        h = F.linear(F.relu(_cbn(w, pre + ".bn_0", net, c)), w[pre + ".fc_0.weight"].squeeze(-1), w[pre + ".fc_0.bias"])
        dx = F.linear(F.relu(_cbn(w, pre + ".bn_1", h, c)), w[pre + ".fc_1.weight"].squeeze(-1), w[pre + ".fc_1.bias"])
        net = net + dx
    out = F.linear(F.relu(_cbn(w, "decoder.bn", net, c)), w["decoder.fc_out.weight"].squeeze(-1), w["decoder.fc_out.bias"])
    return out.squeeze(-1)


# --------------------------------------------------------------------------
# objective + optimiser  (ONet/opt_defense.py:182-239; identical to ConvONet's but for decode(p, z, c))
# --------------------------------------------------------------------------
def losses(w: Weights, p: torch.Tensor, c: torch.Tensor, rep_weight: float, threshold: float = THRESHOLD,
           loss_batch: Optional[int] = None):
    """opt_defense.py:212-225.  Returns (total, occ_loss, rep_loss, logits) with the reference's scaling."""
    lb = float(loss_batch if loss_batch is not None else p.shape[0])
    logits = decode_logits(w, p, c)
    bce = F.binary_cross_entropy_with_logits(logits, torch.full_like(logits, threshold), reduction="none")
    occ = bce.sum() / lb
    rep = p.new_zeros(())
    if rep_weight > 0:
        rep = CO.repulsion_loss(p).sum() / lb * rep_weight
    return occ + rep, occ, rep, logits


def optimize_points(w: Weights, init: torch.Tensor, c: torch.Tensor, rep_weight: float = 500.0,
                    iterations: int = 200, lr: float = 1e-3, threshold: float = THRESHOLD,
                    loss_batch: Optional[int] = None, normalize: bool = True,
                    record: Optional[Sequence[int]] = None):
    """opt_defense.py:182-239: ``iterations + 1`` Adam steps, then unit-sphere normalisation."""
    x = init.clone().float().requires_grad_(True)
    opt = torch.optim.Adam([x], lr=lr)
    snaps = {}
    for i in range(iterations + 1):
        total, _, _, _ = losses(w, x, c, rep_weight, threshold, loss_batch)
        opt.zero_grad()
        total.backward()
        opt.step()
        if record is not None and (i + 1) in record:
            snaps[i + 1] = x.detach().clone()
    out = x.detach()
    if normalize:
        out = CO.normalize_batch_pc(out)
    return (out, snaps) if record is not None else out
