"""PyTorch-CPU float32 restatement of the ConvONet-Opt restoration path.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Functional style: weights
are a plain ``dict[str, torch.Tensor]`` that uses the reference checkpoint's
``state_dict`` key names (``encoder.*`` / ``decoder.*``), so a real
``pretrain/convonet.pth`` can be dropped in unchanged.

Every function names the reference lines it restates (paths relative to
/root/reference).  The op sequence deliberately mirrors the reference
(bmm-kNN + topk, autograd, torch.optim.Adam) so that (i) on CPU the results
are comparable with the imported reference to round-off and (ii) timing this
module is a fair "port" CPU baseline.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Weights = Dict[str, torch.Tensor]

# Resolved hyper-parameters of the shipped config ("R0" in SURVEY.md section 8).
PADDING = 0.1            # ConvONet/configs/default.yaml:30
PLANE_RES = 64           # ConvONet/configs/convonet_3plane_mn40.yaml:22
HIDDEN = 32              # convonet_3plane_mn40.yaml:20,31
C_DIM = 32               # convonet_3plane_mn40.yaml:32
N_BLOCKS = 5             # decoder.py:23 / pointnet.py:33 defaults
THRESHOLD = 0.2          # convonet_3plane_mn40.yaml:46
POINTCLOUD_N = 600       # convonet_3plane_mn40.yaml:7
PLANES = ("xz", "xy", "yz")
PLANE_AXES = {"xz": (0, 2), "xy": (0, 1), "yz": (1, 2)}   # common.py:243-248

REP_NN = 5               # defense/repulsion_loss.py:9
REP_RADIUS = 0.07
REP_H = 0.03
REP_EPS = 1e-12


# --------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------
def _uniform(rng: np.random.Generator, shape, bound: float) -> np.ndarray:
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def make_random_weights(seed: int = 0) -> Dict[str, np.ndarray]:
    """Seeded random weights with the reference architecture and key names.

    Linear layers use the torch default U(-1/sqrt(fan_in), 1/sqrt(fan_in));
    ``fc_1.weight`` is *not* zeroed (the reference zero-initialises it at
    src/layers.py:37, which would make every residual branch constant before
    training - SURVEY.md section 8c); U-Net convs use Xavier-normal weights with
    a small random bias.  numpy ``Generator`` streams are used (not torch's) so
    the fixture generator and the tests draw identical weights.
    """
    rng = np.random.default_rng(seed)
    w: Dict[str, np.ndarray] = {}

    def linear(name: str, n_out: int, n_in: int, bias: bool = True):
        b = 1.0 / math.sqrt(n_in)
        w[name + ".weight"] = _uniform(rng, (n_out, n_in), b)
        if bias:
            w[name + ".bias"] = _uniform(rng, (n_out,), b)

    def conv(name: str, c_out: int, c_in: int, k: int, transpose: bool = False):
        shape = (c_in, c_out, k, k) if transpose else (c_out, c_in, k, k)
        fan_in, fan_out = c_in * k * k, c_out * k * k
        std = math.sqrt(2.0 / (fan_in + fan_out))
        w[name + ".weight"] = (rng.standard_normal(shape) * std).astype(np.float32)
        w[name + ".bias"] = _uniform(rng, (c_out,), 0.05)

    # decoder (LocalDecoder, decoder.py:22-48): 16,001 parameters
    linear("decoder.fc_p", HIDDEN, 3)
    for i in range(N_BLOCKS):
        linear(f"decoder.fc_c.{i}", HIDDEN, C_DIM)
    for i in range(N_BLOCKS):
        linear(f"decoder.blocks.{i}.fc_0", HIDDEN, HIDDEN)
        linear(f"decoder.blocks.{i}.fc_1", HIDDEN, HIDDEN)
    linear("decoder.fc_out", 1, HIDDEN)

    # encoder point-net (LocalPoolPointnet, pointnet.py:31-66)
    linear("encoder.fc_pos", 2 * HIDDEN, 3)
    for i in range(N_BLOCKS):
        linear(f"encoder.blocks.{i}.fc_0", HIDDEN, 2 * HIDDEN)
        linear(f"encoder.blocks.{i}.fc_1", HIDDEN, HIDDEN)
        linear(f"encoder.blocks.{i}.shortcut", HIDDEN, 2 * HIDDEN, bias=False)
    linear("encoder.fc_c", C_DIM, HIDDEN)

    # encoder U-Net (unet.py:140-211), depth 4, start 32, concat, transpose
    chans = [32, 64, 128, 256]
    c_in = C_DIM
    for i, c_out in enumerate(chans):
        conv(f"encoder.unet.down_convs.{i}.conv1", c_out, c_in, 3)
        conv(f"encoder.unet.down_convs.{i}.conv2", c_out, c_out, 3)
        c_in = c_out
    for i in range(3):
        c_out = c_in // 2
        conv(f"encoder.unet.up_convs.{i}.upconv", c_out, c_in, 2, transpose=True)
        conv(f"encoder.unet.up_convs.{i}.conv1", c_out, 2 * c_out, 3)
        conv(f"encoder.unet.up_convs.{i}.conv2", c_out, c_out, 3)
        c_in = c_out
    conv("encoder.unet.conv_final", C_DIM, c_in, 1)
    return w


def to_torch(weights: Dict[str, np.ndarray]) -> Weights:
    return {k: torch.from_numpy(np.ascontiguousarray(v)).float() for k, v in weights.items()}


# --------------------------------------------------------------------------
# coordinates (src/common.py:235-258, 300-315)
# --------------------------------------------------------------------------
def normalize_coordinate(p: torch.Tensor, plane: str, padding: float = PADDING) -> torch.Tensor:
    """[B,T,3] -> [B,T,2] in [0,1): divide by (1+padding+10e-6), shift, clamp.

    The clamp is an *assignment* in the reference (common.py:254-257), i.e. the
    clamped entries carry no gradient; ``torch.where`` with constants keeps
    exactly that behaviour.
    """
    a0, a1 = PLANE_AXES[plane]
    xy = torch.stack((p[..., a0], p[..., a1]), dim=-1)
    xy = xy / (1 + padding + 10e-6) + 0.5
    xy = torch.where(xy >= 1, torch.full_like(xy, 1 - 10e-6), xy)
    xy = torch.where(xy < 0, torch.zeros_like(xy), xy)
    return xy


def coordinate2index(xy: torch.Tensor, reso: int = PLANE_RES) -> torch.Tensor:
    """[B,T,2] -> [B,T] int64 cell = floor(u0*reso) + reso*floor(u1*reso) (common.py:309-311)."""
    ij = (xy * reso).long()
    return ij[..., 0] + reso * ij[..., 1]


# --------------------------------------------------------------------------
# encoder (src/encoder/pointnet.py:104-168, src/layers.py:39-48, unet.py)
# --------------------------------------------------------------------------
def resnet_block_fc(w: Weights, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """ResnetBlockFC.forward (layers.py:39-48)."""
    net = F.linear(F.relu(x), w[prefix + ".fc_0.weight"], w[prefix + ".fc_0.bias"])
    dx = F.linear(F.relu(net), w[prefix + ".fc_1.weight"], w[prefix + ".fc_1.bias"])
    if prefix + ".shortcut.weight" in w:
        x = F.linear(x, w[prefix + ".shortcut.weight"])
    return x + dx


def scatter_max_gather(feat: torch.Tensor, index: torch.Tensor, n_cells: int) -> torch.Tensor:
    """Per-cell channel-wise max gathered back to the points.

    Restates ``scatter_max(...)[0].gather(...)`` of pool_local
    (pointnet.py:115-120; torch-scatter 2.0.5, not vendored).  feat [B,T,C],
    index [B,T] -> [B,T,C].  Only occupied cells are ever read back, so the
    fill value of empty cells is irrelevant.
    """
    B, T, C = feat.shape
    idx = index[:, :, None].expand(B, T, C)
    grid = feat.new_full((B, n_cells, C), -float("inf"))
    grid = grid.scatter_reduce(1, idx, feat, reduce="amax", include_self=True)
    return grid.gather(1, idx)


def scatter_mean_plane(c: torch.Tensor, index: torch.Tensor, reso: int = PLANE_RES) -> torch.Tensor:
    """scatter_mean into a zero plane, reshaped [B,C,reso,reso] (pointnet.py:75-80)."""
    B, T, C = c.shape
    idx = index[:, :, None].expand(B, T, C)
    summed = c.new_zeros(B, reso * reso, C).scatter_add(1, idx, c)
    count = c.new_zeros(B, reso * reso).scatter_add(1, index, torch.ones_like(index, dtype=c.dtype))
    mean = summed / count.clamp(min=1)[:, :, None]
    return mean.permute(0, 2, 1).reshape(B, C, reso, reso)


def pointnet_features(w: Weights, p: torch.Tensor, return_stages: bool = False):
    """Point-wise part of LocalPoolPointnet.forward (pointnet.py:124-156): [B,T,3] -> c [B,T,32]."""
    coords = {pl: normalize_coordinate(p, pl) for pl in PLANES}
    index = {pl: coordinate2index(coords[pl]) for pl in PLANES}
    stages: List[torch.Tensor] = []
    pooled_stages: List[torch.Tensor] = []
    net = F.linear(p, w["encoder.fc_pos.weight"], w["encoder.fc_pos.bias"])
    net = resnet_block_fc(w, "encoder.blocks.0", net)
    stages.append(net)
    for i in range(1, N_BLOCKS):
        pooled = sum(scatter_max_gather(net, index[pl], PLANE_RES ** 2) for pl in PLANES)      # pool_local, pointnet.py:104-122
        pooled_stages.append(pooled)
        net = resnet_block_fc(w, f"encoder.blocks.{i}", torch.cat([net, pooled], dim=2))
        stages.append(net)
    c = F.linear(net, w["encoder.fc_c.weight"], w["encoder.fc_c.bias"])
    if return_stages == "pooled":
        return c, index, stages, pooled_stages
    if return_stages:
        return c, index, stages
    return c, index


def unet_forward(w: Weights, x: torch.Tensor, prefix: str = "encoder.unet") -> torch.Tensor:
    """UNet.forward (unet.py:225-239) for depth 4 / concat / transpose up-conv."""
    def conv(name, t, pad):
        return F.conv2d(t, w[f"{prefix}.{name}.weight"], w[f"{prefix}.{name}.bias"], padding=pad)

    skips = []
    for i in range(4):
        x = F.relu(conv(f"down_convs.{i}.conv1", x, 1))
        x = F.relu(conv(f"down_convs.{i}.conv2", x, 1))
        skips.append(x)
        if i < 3:
            x = F.max_pool2d(x, 2, 2)
    for i in range(3):
        up = F.conv_transpose2d(x, w[f"{prefix}.up_convs.{i}.upconv.weight"],
                                w[f"{prefix}.up_convs.{i}.upconv.bias"], stride=2)
        x = torch.cat((up, skips[-(i + 2)]), dim=1)
        x = F.relu(conv(f"up_convs.{i}.conv1", x, 1))
        x = F.relu(conv(f"up_convs.{i}.conv2", x, 1))
    return conv("conv_final", x, 0)


def encode_inputs(w: Weights, sel: torch.Tensor, return_pre_unet: bool = False):
    """``generator.model.encode_inputs`` (models/__init__.py:52 -> pointnet.py:124-168).

    sel [B,T,3] -> {'xz','xy','yz': [B,32,64,64]}.
    """
    c, index = pointnet_features(w, sel)
    pre = {pl: scatter_mean_plane(c, index[pl]) for pl in PLANES}
    out = {pl: unet_forward(w, pre[pl]) for pl in PLANES}
    if return_pre_unet:
        return out, pre
    return out


# --------------------------------------------------------------------------
# decoder (src/conv_onet/models/decoder.py:50-95)
# --------------------------------------------------------------------------
def sample_plane_feature(p: torch.Tensor, plane_feat: torch.Tensor, plane: str) -> torch.Tensor:
    """decoder.py:50-57: bilinear / border / align_corners grid_sample -> [B,C,K]."""
    xy = normalize_coordinate(p, plane)
    vgrid = 2.0 * xy[:, :, None] - 1.0
    return F.grid_sample(plane_feat, vgrid, padding_mode="border", align_corners=True,
                         mode="bilinear").squeeze(-1)


def decode_logits(w: Weights, p: torch.Tensor, planes: Dict[str, torch.Tensor]) -> torch.Tensor:
    """``generator.model.decode(p, c).logits`` (decoder.py:69-95): [B,K,3] -> [B,K]."""
    c = sum(sample_plane_feature(p, planes[pl], pl) for pl in PLANES).transpose(1, 2)
    net = F.linear(p, w["decoder.fc_p.weight"], w["decoder.fc_p.bias"])
    for i in range(N_BLOCKS):
        net = net + F.linear(c, w[f"decoder.fc_c.{i}.weight"], w[f"decoder.fc_c.{i}.bias"])
        net = resnet_block_fc(w, f"decoder.blocks.{i}", net)
    out = F.linear(F.relu(net), w["decoder.fc_out.weight"], w["decoder.fc_out.bias"])
    return out.squeeze(-1)


# --------------------------------------------------------------------------
# repulsion loss (defense/repulsion_loss.py:18-54, defense/pn_utils.py:64-83)
# --------------------------------------------------------------------------
def knn_point(k: int, points: torch.Tensor) -> torch.Tensor:
    """pn_utils.py:64-83: expanded-form squared distances, top-(k+1), drop column 0."""
    pc = points.detach()
    inner = -2.0 * torch.matmul(pc, pc.transpose(2, 1))
    xx = torch.sum(pc ** 2, dim=2, keepdim=True)
    dist = xx.transpose(2, 1) + inner + xx
    _, top_idx = (-dist).topk(k=k + 1, dim=-1)
    return top_idx[:, :, 1:]


def repulsion_loss(pred: torch.Tensor, idx: Optional[torch.Tensor] = None) -> torch.Tensor:
    """repulsion_loss.py:43-54: [B,K,3] -> [B] (mean over K*5 of (r-d)*exp(-d^2/h^2))."""
    if idx is None:
        idx = knn_point(REP_NN, pred)
    B, K, k = idx.shape
    nbr = pred[torch.arange(B)[:, None, None], idx]          # index_points, pn_utils.py:6-23
    diff = nbr - pred[:, :, None, :]
    dist2 = torch.clamp(torch.sum(diff ** 2, dim=-1), min=REP_EPS)
    dist = torch.sqrt(dist2)
    weight = torch.exp(-((dist / REP_H) ** 2))
    return torch.mean((REP_RADIUS - dist) * weight, dim=[1, 2])


# --------------------------------------------------------------------------
# SOR (defense/SOR.py:22-49)
# --------------------------------------------------------------------------
def sor_keep_mask(x: torch.Tensor, k: int = 2, alpha: float = 1.1) -> Tuple[torch.Tensor, torch.Tensor]:
    """[B,K,3] f32 -> (keep mask [B,K] bool, value [B,K] f64)."""
    pc = x.detach().double()
    inner = -2.0 * torch.matmul(pc, pc.transpose(2, 1))
    xx = torch.sum(pc ** 2, dim=2, keepdim=True)
    dist = xx.transpose(2, 1) + inner + xx
    neg_value, _ = (-dist).topk(k=k + 1, dim=-1)
    value = torch.mean(-(neg_value[..., 1:]), dim=-1)
    threshold = torch.mean(value, dim=-1) + alpha * torch.std(value, dim=-1)
    return value <= threshold[:, None], value


# --------------------------------------------------------------------------
# driver pieces (ConvONet/opt_defense.py)
# --------------------------------------------------------------------------
def preprocess_pc(pc: np.ndarray, padding_scale: float = 0.9) -> np.ndarray:
    """opt_defense.py:122-127: centre, divide by the largest bbox extent, scale (numpy f32)."""
    pc = np.asarray(pc, dtype=np.float32)
    centered = pc - np.mean(pc, axis=0)
    scale = (np.max(centered, axis=0) - np.min(centered, axis=0)).max()
    return (centered / scale * padding_scale).astype(np.float32)


def init_points(all_pc: Sequence[np.ndarray], idx: np.ndarray, noise: np.ndarray,
                init_sigma: float = 0.01, padding_scale: float = 0.9) -> torch.Tensor:
    """opt_defense.py:149-179 with the random draws passed in (idx [B,n], noise [B,n,3] ~ N(0,1))."""
    pts = np.stack([np.asarray(all_pc[b], np.float32)[idx[b]] for b in range(len(all_pc))])
    pts = torch.from_numpy(pts) + torch.from_numpy(noise.astype(np.float32)) * init_sigma
    return torch.clamp(pts, min=-0.5 * padding_scale, max=0.5 * padding_scale)


def normalize_batch_pc(points: torch.Tensor) -> torch.Tensor:
    """opt_defense.py:76-83: centre on the centroid, divide by the largest norm."""
    points = points - torch.mean(points, dim=1, keepdim=True)
    dist = torch.sum(points ** 2, dim=2) ** 0.5
    return points / torch.max(dist, dim=1)[0][:, None, None]


def losses(w: Weights, p: torch.Tensor, planes: Dict[str, torch.Tensor], rep_weight: float,
           threshold: float = THRESHOLD, loss_batch: Optional[int] = None):
    """One evaluation of the objective (opt_defense.py:212-225).

    ``loss_batch`` is the number of clouds the reference averages over (the 1/B
    factor of both ``torch.mean`` calls); it defaults to ``p.shape[0]``.
    Returns (total, occ_loss, rep_loss, logits) with the reference's scaling.
    """
    B, K = p.shape[:2]
    lb = float(loss_batch if loss_batch is not None else B)
    logits = decode_logits(w, p, planes)
    target = torch.full_like(logits, threshold)
    bce = F.binary_cross_entropy_with_logits(logits, target, reduction="none")
    occ = bce.sum() / lb                      # == mean over [B,K] * K when lb == B
    rep = p.new_zeros(())
    if rep_weight > 0:
        rep = repulsion_loss(p).sum() / lb * rep_weight
    return occ + rep, occ, rep, logits


def optimize_points(w: Weights, init: torch.Tensor, planes: Dict[str, torch.Tensor],
                    rep_weight: float = 500.0, iterations: int = 200, lr: float = 1e-3,
                    threshold: float = THRESHOLD, loss_batch: Optional[int] = None,
                    normalize: bool = True, record: Optional[Sequence[int]] = None):
    """opt_defense.py:182-239: ``iterations + 1`` Adam steps, then unit-sphere normalisation.

    Returns the final points (torch [B,K,3]); with ``record`` (step counts) also
    a dict {n_steps: points after n_steps updates, un-normalised}.
    """
    x = init.clone().float().requires_grad_(True)
    opt = torch.optim.Adam([x], lr=lr)
    snaps = {}
    for i in range(iterations + 1):
        total, _, _, _ = losses(w, x, planes, rep_weight, threshold, loss_batch)
        opt.zero_grad()
        total.backward()
        opt.step()
        if record is not None and (i + 1) in record:
            snaps[i + 1] = x.detach().clone()
    out = x.detach()
    if normalize:
        out = normalize_batch_pc(out)
    if record is not None:
        return out, snaps
    return out


def adam_step(x: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, t: int,
              lr: float = 1e-3, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8):
    """torch.optim.Adam single-tensor update (torch >= 2: lerp form); t is the 1-based step."""
    m = m + (g - m) * (1 - b1)
    v = v * b2 + (1 - b2) * g * g
    step_size = lr / (1 - b1 ** t)
    denom = v.sqrt() / math.sqrt(1 - b2 ** t) + eps
    return x - step_size * (m / denom), m, v
