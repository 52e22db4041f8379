"""Multi-GPU host logic: probe-column sharding (design 1).

One process per GPU (``torch.distributed``; backend "nccl" == RCCL over xGMI on ROCm, "gloo" in the
CPU tests).  X is replicated, the probe vectors are partitioned over ranks, and because every mBCG
quantity (alpha, beta, rho, |r|, tridiagonals) is per column, ranks never exchange vectors.  The only
data-path collectives are
  * one all-reduce of two floats per CG iteration (sum of residual norms, column count) so that the
    reference's GLOBAL mean-residual stopping rule fires at the same iteration on every rank, and
  * one scalar all-reduce of the SLQ partial sums (and one of the d+2 hyper-parameter gradients).
The reference's only multi-GPU code, ``MultiDeviceKernel`` (``gpytorch/kernels/multi_device_kernel.py:
49-92``: single-process DataParallel row-chunking with peer copies of V every matmul), is what this
replaces.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return None
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend, rank=int(os.environ["RANK"]), world_size=world)
    return dist.group.WORLD


def probe_shard(t_total: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous, balanced partition of t_total probe columns: the first (t_total % world) ranks get
    one extra column.  Returns (start, stop)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(t_total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def allreduce_residual_stats(sum_rnorm: torch.Tensor, count: torch.Tensor, group=None):
    """Global mean relative residual from per-rank (sum of norms, column count)."""
    pack = torch.stack([sum_rnorm.reshape(()).to(torch.float64), count.reshape(()).to(torch.float64)])
    if group is not None or (dist.is_initialized() and dist.get_world_size() > 1):
        pack = pack.to(sum_rnorm.device)
        dist.all_reduce(pack, group=group)
    return pack[0] / pack[1]


def allreduce_sum_(t: torch.Tensor, group=None):
    """In-place sum all-reduce.  RCCL ("nccl") reduces device tensors directly over xGMI; the gloo backend
    (CPU tests, or several ranks sharing one GPU) is given a host copy."""
    if group is not None or (dist.is_initialized() and dist.get_world_size() > 1):
        if t.is_cuda and dist.get_backend(group) == "gloo":
            tmp = t.detach().cpu()
            dist.all_reduce(tmp, group=group)
            t.copy_(tmp)
        else:
            dist.all_reduce(t, group=group)
    return t
