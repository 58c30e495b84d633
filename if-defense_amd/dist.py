"""Multi-GPU: one process per GPU, shape batches sharded contiguously, ONE all-gather at the end.

The reference's hot path is single-device (ConvONet/command.txt:3 pins CUDA_VISIBLE_DEVICES=0); its only
multi-GPU code shards the *attacks* with DistributedSampler and merges per-rank .npz files offline
(baselines/attack_scripts/targeted_knn_attack.py:97-174, util/merge_attack_results.py:7-51).  Here every
cloud is independent from SOR to normalisation, so ranks take contiguous ranges and the restored array is
re-assembled with a single collective (RCCL over xGMI with backend 'nccl'; 'gloo' for the CPU tests).
Nothing is exchanged inside the 501-step loop.
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int, int]:
    """Contiguous range [lo, hi) of rank `rank` and the padded per-rank length ceil(n / world)."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per), per


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).  Returns
    (rank, world, local_rank).  A plain `python x.py` run (WORLD_SIZE unset) stays single-process with no process
    group; under torchrun the group is created even for one rank, so that `--nproc-per-node 1` on a one-GPU box
    exercises the same RCCL calls (communicator, barrier, all-gather) as the N > 1 runs."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # RCCL shares buffers between the ranks' processes through HIP IPC handles; the hosts of this build only support the
    # dmabuf flavour (HSA_ENABLE_IPC_MODE_LEGACY=0; with the legacy mode hipIpcGetMemHandle fails with "invalid argument").
    # It must be in the environment before the first HIP call of the process.  Only a default: export the variable
    # yourself (any value) to override it on hosts that need the legacy mode.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "WORLD_SIZE" in os.environ and "MASTER_ADDR" in os.environ and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            n_dev = torch.cuda.device_count()
            if local >= n_dev:
                raise RuntimeError(
                    "LOCAL_RANK=%d but this node shows %d GPU(s) to the process (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES=%s): "
                    "one process per GPU - launch with --nproc-per-node <= %d, or make more devices visible" %
                    (local, n_dev, os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES", "unset")), max(n_dev, 1)))
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)      # bind the RCCL communicator (and barrier()) to this rank's GPU
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


class AgreedFailure(RuntimeError):
    """Raised on EVERY rank once an agreement point (`agree`) has reported a failure somewhere.  After it no rank
    enters another collective: the ranks stop together.  `own` is this rank's exception (None on the healthy ranks)."""

    def __init__(self, own=None):
        super().__init__("a rank failed" if own is None else "%s: %s" % (type(own).__name__, own))
        self.own = own


def any_rank_failed(failed: bool, device=None) -> bool:
    """One MAX all-reduce of a status flag: True on every rank if any rank reports a failure (single process: `failed`)."""
    if not dist.is_initialized():
        return bool(failed)
    if dist.get_backend() == "nccl":            # RCCL reduces device tensors only
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    else:
        dev = torch.device("cpu")
    flag = torch.tensor([1.0 if failed else 0.0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    return bool(flag.item() > 0)


def agree(err, device=None):
    """Agreement point of the sharded drivers: every rank reports whether its local work since the last agreement
    failed (`err`: the exception, or None).  If any rank failed, AgreedFailure is raised on all of them - BEFORE the next
    data collective, so a rank that threw during its compute never leaves its peers waiting in an all-gather (mismatched
    collectives hang on RCCL).  The protocol the drivers keep: exactly one agreement reports a failure (a rank that fails
    anywhere calls the next agreement with its error), and nobody calls a collective after an agreement has failed."""
    if any_rank_failed(err is not None, device):
        raise AgreedFailure(err)


def shutdown():
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def gather_shards(local: torch.Tensor, n_total: int, per: int) -> torch.Tensor:
    """All-gather equal-length (padded) shards and trim: every rank returns the full [n_total, ...] array."""
    if not dist.is_initialized():
        return local[:n_total]
    world = dist.get_world_size()
    # (gloo moves host memory: device shards - several processes sharing ONE GPU in tests/test_gpu_two_processes.py - go through the host)
    via_host = dist.get_backend() != "nccl" and local.is_cuda
    src = local.cpu() if via_host else local
    pad = torch.zeros((per,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    pad[:src.shape[0]] = src
    out = torch.empty((world * per,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, pad)          # RCCL ncclAllGather over xGMI (one call, ~30 MB for 2468 clouds)
    return (out.to(local.device) if via_host else out)[:n_total]


def defend_sharded(defend: Callable[[np.ndarray, int, int], torch.Tensor], pc: np.ndarray, device=None) -> torch.Tensor:
    """Run `defend(shard, cloud_index_base, total)` on this rank's contiguous shard of `pc`, agree that every rank's
    compute succeeded (`agree`: raises AgreedFailure everywhere if one did not), then all-gather.

    Because random draws are keyed by the global cloud index and the 1/B loss factor by the reference batch a
    cloud belongs to, the gathered array is bit-identical to a single-process run (tests/test_host_cpu.py,
    tests/test_gpu_parity.py::test_defend_point_cloud_end_to_end_and_sharding).
    """
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    n = len(pc)
    lo, hi, per = shard_range(n, rank, world)
    local, err = None, None
    try:
        local = defend(pc[lo:hi], lo, n)
    except Exception as e:                      # noqa: BLE001  (reported to every rank by the agreement below)
        err = e
    agree(err, device)
    return gather_shards(local, n, per)
