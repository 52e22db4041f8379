"""Multi-GPU host logic: probe-column sharding (design 1) and row sharding for the small-t
solves (design 2, :class:`RowShard`).

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


def allreduce_max_int(v: int, group=None) -> int:
    """max over ranks of a host integer (one tiny collective; used to make launch / polling schedules rank-independent)."""
    if group is None and not (dist.is_initialized() and dist.get_world_size() > 1):
        return v
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.tensor([v], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def broadcast_(t: torch.Tensor, src_group_rank: int, group=None):
    """In-place broadcast from rank ``src_group_rank`` OF THE GROUP (host-staged under gloo with device tensors)."""
    if group is None and not (dist.is_initialized() and dist.get_world_size() > 1):
        return t
    src = dist.get_global_rank(group, src_group_rank) if (group is not None and group is not dist.group.WORLD) else src_group_rank
    if t.is_cuda and dist.get_backend(group) == "gloo":
        tmp = t.detach().cpu()
        dist.broadcast(tmp, src=src, group=group)
        t.copy_(tmp)
    else:
        dist.broadcast(t, src=src, group=group)
    return t


class RowShard:
    """Row sharding of K_hat for solves with few right-hand sides.

    Rank g owns the contiguous rows ``[g * n_pad, min(n, (g+1) * n_pad))`` of K_hat (``n_pad`` = ceil(n / G) rounded up
    to 4); X is replicated.  Per product: one all-gather of the t local vector slices (n * t * 4 bytes in total;
    t = 1 for the predictive-mean CG and Lanczos) followed by a RECTANGULAR fused K*V (local rows x all columns: 1/G of
    the kernel-generation work, which is what bounds these products).  Per CG iteration: two all-reduces of t partial
    sums (d^T q; r^T r).  The padded tail ``[n, G * n_pad)`` of the gathered vectors is identically zero, so the rows
    appended to the replicated cloud for it (copies of the last point) never contribute."""

    def __init__(self, xp_full, group):
        from . import backend as B

        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        n = xp_full.n
        self.n = n
        self.n_pad = B.round_up((n + self.world - 1) // self.world, 4)
        self.r0 = self.rank * self.n_pad
        self.r1 = min(n, self.r0 + self.n_pad)
        if (self.world - 1) * self.n_pad >= n:
            raise ValueError(f"row sharding needs at least one row per rank (n = {n}, world = {self.world})")
        self.n_loc = self.r1 - self.r0
        self.np_all = self.world * self.n_pad
        xp = xp_full.xp
        self.x_loc = B.PreparedPoints(xp[self.r0 : self.r1], self.n_loc, xp_full.d, xp_full.dp, xp_full.kind, xp_full.param)
        if self.np_all > n:
            xp = torch.cat([xp, xp[-1:].expand(self.np_all - n, xp.shape[1])], dim=0).contiguous()
        self.x_all = B.PreparedPoints(xp, self.np_all, xp_full.d, xp_full.dp, xp_full.kind, xp_full.param)
        self.x_all._zmax2 = xp_full.zmax2
        self.x_loc._zmax2 = xp_full.zmax2

    # ---- vectors: probe-major [t, ld] ------------------------------------------------------------------------------
    def local(self, full_t: torch.Tensor) -> torch.Tensor:
        """[t, >= n] -> this rank's slice as a fresh probe-major [t, round_up(n_loc, 4)] tensor."""
        from . import backend as B

        out = torch.zeros(full_t.shape[0], B.round_up(self.n_loc, 4), device=full_t.device, dtype=full_t.dtype)
        out[:, : self.n_loc] = full_t[:, self.r0 : self.r1]
        return out

    def gather(self, loc_t: torch.Tensor) -> torch.Tensor:
        """Local slices [t, >= n_loc] of every rank -> [t, G * n_pad] (zero in the padded tail of every slice)."""
        t = loc_t.shape[0]
        send = torch.zeros(t, self.n_pad, device=loc_t.device, dtype=loc_t.dtype)
        send[:, : self.n_loc] = loc_t[:, : self.n_loc]
        recv = torch.empty(self.world * t, self.n_pad, device=loc_t.device, dtype=loc_t.dtype)  # rank-major concatenation
        if loc_t.is_cuda and dist.get_backend(self.group) == "gloo":
            r = recv.cpu()
            dist.all_gather_into_tensor(r, send.cpu(), group=self.group)
            recv.copy_(r)
        else:
            dist.all_gather_into_tensor(recv, send, group=self.group)
        if t == 1:
            return recv.reshape(1, self.np_all)
        return recv.view(self.world, t, self.n_pad).permute(1, 0, 2).reshape(t, self.np_all).contiguous()

    def kv_local(self, loc_t: torch.Tensor, scale=None, dscale=None, dvec_loc=None) -> torch.Tensor:
        """This rank's rows of (scale * K + dscale * I + diag(dvec)) @ V for V given by its local slices."""
        from . import backend as B

        full = self.gather(loc_t)
        return B.kv(self.x_loc, self.x_all, full, scale=scale, dscale=dscale, vd=loc_t if dscale is not None else None, dvec=dvec_loc)

    def allreduce_partials(self, fscratch: torch.Tensor, offset: int, t: int, stride: int):
        """Sum over ranks of one of the solver's per-column partial arrays (include/gpamd.h, gpamd_cg_partials_layout):
        collapse each column's partials, all-reduce the t sums, store them in entry 0 and clear the rest."""
        v = fscratch[offset : offset + t * stride].view(t, stride)
        sums = v.sum(dim=1)
        allreduce_sum_(sums, self.group)
        v.zero_()
        v[:, 0] = sums

    def allreduce(self, x: torch.Tensor) -> torch.Tensor:
        return allreduce_sum_(x, self.group)

    def broadcast(self, x: torch.Tensor) -> torch.Tensor:
        """In place, from the first rank of the group."""
        src = dist.get_global_rank(self.group, 0) if self.group is not dist.group.WORLD else 0
        if x.is_cuda and dist.get_backend(self.group) == "gloo":
            tmp = x.detach().cpu()
            dist.broadcast(tmp, src=src, group=self.group)
            x.copy_(tmp)
        else:
            dist.broadcast(x, src=src, group=self.group)
        return x


# ======================================================================================================================================
# Layout policy: how many probe shares x row blocks for one MLL evaluation on G ranks (replaces the "which devices" argument of the
# reference's MultiDeviceKernel(base_kernel, device_ids), kernels/multi_device_kernel.py:24-47 -- the user names no layout there either).
#
# Measured column ladder of the fused K*V on ONE MI355X, per 2.5e11 pairs (n = m = 500 000; profiles/r05_s8_bench_kernel_stats.csv,
# r02_s38_scale_check_split_default.json): kernel generation is replicated on every probe share, so a launch costs
# max(generation, 32-column contraction tiles) and is linear in rows x columns-of-K:
KV_LADDER_MS = {
    # contraction: (<= 4 columns: first + each further, one 32-column tile, each further tile, extra column riding on the VALU: beside ONE tile / beside more)
    "split": (18.6, 5.2, 45.0, 44.0, 13.5, 2.0),     # hi/lo-split f16 operands (library default): 1: 18.6, 2: 23.8, 9: 45, 33: 58.5, 65: 91
    "f32": (18.6, 5.2, 121.0, 118.0, 10.0, 3.0),     # fp32 MFMA (bench.py --contraction f32): 33: 131 (C4 share 519.8 / 4), 65: 242
}
XGMI_LINK_GBS = 100.0     # effective bytes/s of one ring step over one xGMI link (153 GB/s peak per link; ring collectives are per-link bound)
ALLREDUCE_SMALL_MS = 0.05  # one stream-ordered all-reduce of a few floats


def kv_cost_ms(cols: int, rows: int, m: int, contraction: str = "split") -> float:
    """Modelled time of ONE fused K*V launch with ``cols`` right-hand sides on ``rows`` x ``m`` kernel entries (the ladder above, scaled by pairs)."""
    first, nxt, tile, more, extra1, extra = KV_LADDER_MS[contraction]
    if cols <= 4:
        base = first + nxt * (cols - 1)
    else:
        ex = 1 if cols % 32 == 1 and cols > 32 else 0
        tiles = (cols - ex + 31) // 32
        base = tile + more * (tiles - 1) + (extra1 if tiles == 1 else extra) * ex
    return base * (float(rows) * float(m)) / 2.5e11


def grid_cost_ms(P: int, R: int, n: int, t_total: int, contraction: str = "split", rhs_cols: int = 1) -> float:
    """Modelled time of one mBCG iteration of the MLL solve on a P x R grid: the widest probe share's launch on its row block + (R > 1) one
    all-gather of the search directions over the row group (ring: (R - 1) / R of 4 n cols bytes through one link) and the two small
    all-reduces of the solver's inner products."""
    cols = -(-t_total // P) + rhs_cols            # the first share also carries the rhs column(s)
    rows = -(-n // R)
    ms = kv_cost_ms(cols, rows, n, contraction) + ALLREDUCE_SMALL_MS
    if R > 1:
        ms += 4.0 * n * cols * (R - 1) / R / (XGMI_LINK_GBS * 1e9) * 1e3 + 2 * ALLREDUCE_SMALL_MS
    return ms


def choose_grid(world: int, n: int, t_total: int, contraction: str = "split", rhs_cols: int = 1, allow_rows: bool = True) -> tuple[int, int]:
    """(P probe shares, R row blocks), P * R == world, minimising :func:`grid_cost_ms`; ties go to the layout with fewer row blocks (no
    all-gather).  ``allow_rows=False``: operators whose rows cannot be sharded (structured operators) -> (world, 1).  Every share needs at
    least one probe and every row block at least one 512-row group of the Gram-form kernels."""
    best = (world, 1)
    if not allow_rows:
        return best
    best_ms = None
    for R in range(1, world + 1):
        if world % R:
            continue
        P = world // R
        if P > max(t_total, 1) or (R > 1 and n // R < 512):
            continue
        ms = grid_cost_ms(P, R, n, t_total, contraction, rhs_cols)
        if best_ms is None or ms < best_ms * (1.0 - 1e-9):
            best, best_ms = (P, R), ms
    return best


_GRID_GROUPS: dict = {}


def grid_groups(P: int, R: int, base=None):
    """The two subgroups of THIS rank on a P x R grid of the ranks of ``base`` (default WORLD), rank = p * R + r: (probe group = the P ranks with
    its r -- stopping rule, SLQ sums --, row group = the R ranks with its p -- all-gather of the search directions, inner products); ``None``
    for a dimension of extent 1.  ``new_group`` is collective: every rank creates every group, in the same order; the result is cached per
    (base, P, R), so repeated evaluations create nothing."""
    world = dist.get_world_size(base)
    if P * R != world:
        raise ValueError(f"grid {P}x{R} needs {P * R} ranks, the group has {world}")
    key = (id(base) if base is not None else 0, P, R)
    if key not in _GRID_GROUPS:
        rank = dist.get_rank(base)
        ranks = list(range(world)) if base is None or base is dist.group.WORLD else dist.get_process_group_ranks(base)
        pgs = [dist.new_group([ranks[p * R + r] for p in range(P)]) for r in range(R)] if (P > 1 and R > 1) else None
        rgs = [dist.new_group([ranks[p * R + r] for r in range(R)]) for p in range(P)] if (P > 1 and R > 1) else None
        whole = base if base is not None else dist.group.WORLD
        if R == 1:
            _GRID_GROUPS[key] = (whole if P > 1 else None, None)
        elif P == 1:
            _GRID_GROUPS[key] = (None, whole)
        else:
            _GRID_GROUPS[key] = (pgs[rank % R], rgs[rank // R])
    return _GRID_GROUPS[key]
