"""BBMM building blocks on top of the HIP kernels: preconditioner, probe vectors, stochastic
Lanczos quadrature, and the ``inv_quad_logdet`` / ``solve`` drivers for K_hat = theta*K + sigma^2 I.

Mirrors (third-party linear_operator v0.6.x; restated in oracle/ and SURVEY.md Appendix A):
  * ``AddedDiagLinearOperator._preconditioner / _init_cache``      (A.4)
  * ``LinearOperator._probe_vectors_and_norms``                     (A.5)
  * ``InvQuadLogdet.forward`` + ``StochasticLQ`` + ``lanczos_tridiag_to_diag``   (A.6)
Reference consumers: ``gpytorch/distributions/multivariate_normal.py:249-251`` and
``gpytorch/models/exact_prediction_strategies.py:286``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

from . import backend as B
from .distributed import allreduce_sum_, broadcast_
from . import settings
from .linear_cg import CGInfo, Preconditioner, linear_cg


def build_preconditioner(x: B.PreparedPoints, scale, sigma2: torch.Tensor, rank=None, tol=None, min_size=None, dvec=None):
    """Pivoted-Cholesky preconditioner P = L L^T + s2 I (constant-diagonal branch of A.4); with ``dvec`` (a fixed per-point noise
    vector [>= n] added on top of s2: FixedNoiseGaussianLikelihood) the non-constant-diagonal branch, P = L L^T + diag(s2 + dvec).

    Returns ``None`` when disabled (rank 0 or n < min_preconditioning_size), else a
    :class:`Preconditioner` holding Q1 (as [k, ld] rows), log|P| and L^T.

    The reference takes a Householder QR of [L; sqrt(s2) I]; here the same thin factor is obtained
    with two rounds of Cholesky-QR (GEMM + k x k Cholesky in float64 + triangular solve), which
    keeps the n x k work on rocBLAS GEMMs.  Q1 Q1^T and |diag R| -- the only quantities used -- are
    identical up to rounding."""
    n = x.n
    rank = settings.max_preconditioner_size.resolve(n) if rank is None else rank
    tol = settings.preconditioner_tolerance.value() if tol is None else tol
    min_size = settings.min_preconditioning_size.value() if min_size is None else min_size
    if rank == 0 or n < min_size:
        return None
    s2v = float(sigma2.detach().reshape(-1)[0])
    if (s2v <= 0.0 and dvec is None) or (dvec is not None and not bool((dvec.detach()[:n] + s2v > 0).all())):
        return None  # (P = L L^T + D is singular without a positive diagonal)
    lt, _, k = B.pivoted_cholesky(x, scale, rank, tol)  # [k, n]
    if not bool(torch.isfinite(lt).all()):
        import warnings

        from .linear_cg import NumericalWarning

        warnings.warn("NaNs encountered in preconditioner computation. Attempting to continue without preconditioning.", NumericalWarning)
        return None
    if dvec is not None:
        return preconditioner_from_factor(lt, n, sigma2.detach().reshape(()).to(dvec.dtype) + dvec.detach()[:n], x.dtype, noise_is_vector=True)
    return preconditioner_from_factor(lt, n, sigma2, x.dtype)


def gram_rows(rows: torch.Tensor) -> torch.Tensor:
    """rows rows^T in float64 for tall-skinny probe-major ``rows`` [k, n] (float32 or float64, unit column stride): the library's reduction kernel
    on the device, a torch GEMM elsewhere."""
    k, n = rows.shape
    if not rows.is_cuda or rows.dtype not in (torch.float32, torch.float64) or rows.stride(1) != 1:
        r64 = rows.to(torch.float64)
        return r64 @ r64.t()
    from ._lib import check, lib

    L = lib()
    G = torch.empty(k, k, device=rows.device, dtype=torch.float64)
    ws = torch.empty(max(int(L.gpamd_precond_coef_workspace_doubles(n, min(k, 16), k)), 1), device=rows.device, dtype=torch.float64)
    fn = L.gpamd_block_project_f32 if rows.dtype == torch.float32 else L.gpamd_block_project_f64
    check(fn(B._ptr(rows), rows.stride(0), k, B._ptr(rows), rows.stride(0), k, n, B._ptr(G), B._ptr(ws), ws.numel(), B._stream(rows.device)), "block_project")
    return 0.5 * (G + G.t())


def preconditioner_from_factor(lt: torch.Tensor, n: int, noise: torch.Tensor, wd, noise_is_vector: bool = False) -> Preconditioner:
    """P = L L^T + D from the (pivoted-Cholesky) factor ``lt`` [k, >= n] and the diagonal D: a scalar sigma^2 (A.4's constant branch)
    or a vector [>= n] (the reference's ``_init_cache_for_non_constant_diag``: fixed heteroskedastic noise, per-task noise).

    The reference takes a Householder QR of [L; sqrt(s2) I]; here the same thin factor comes from two rounds of Cholesky-QR in
    float64 (n x k GEMMs + k x k Cholesky + triangular solve).  For a vector D the system is rescaled: Lt = D^-1/2 L, sigma^2 = 1,
    log|P| = sum log d + log|Lt^T Lt + I| and the apply is D^-1/2 (I - Q1 Q1^T) D^-1/2 (``Preconditioner.dinv_sqrt``)."""
    dev = lt.device
    k = lt.shape[0]
    ld = B.round_up(n, 4)
    eye = torch.eye(k, device=dev, dtype=torch.float64)
    ltd = lt[:, :n].to(torch.float64)
    extra_logdet = 0.0
    dinv_sqrt = None
    if noise_is_vector:
        d = noise.detach().reshape(-1)[:n].to(torch.float64)
        di = d.rsqrt()
        ltd = ltd * di.unsqueeze(0)
        extra_logdet = d.log().sum()
        s2 = torch.ones((), device=dev, dtype=torch.float64)
        dinv_sqrt = torch.zeros(ld, device=dev, dtype=torch.float64)
        dinv_sqrt[:n] = di
    else:
        s2 = noise.detach().reshape(()).to(torch.float64)
    # Cholesky-QR twice, all in float64.  The two k x n x k Gram products go through the library's own reduction kernel on the device (float64
    # accumulation, gpamd_block_project_*): rocBLAS' float64 GEMM takes 35 ms for k = 15, n = 217 437 and 5.9 ms for k = 100, n = 36 584 -- 70 / 12 ms
    # per MLL evaluation of the reference's road3d / protein workloads (profiles/r05_s1_workload_*_kernel_stats.csv: "Cijk_Alik_Bljk_DB_...")
    r1 = torch.linalg.cholesky(gram_rows(lt[:, :n] if not noise_is_vector else ltd) + s2 * eye, upper=True)   # G = L^T L + s2 I = R1^T R1
    r1inv = torch.linalg.solve_triangular(r1, eye, upper=True)
    q1t = r1inv.t() @ ltd                                                       # [k, n] = (L R1^-1)^T
    g2 = gram_rows(q1t) + s2 * (r1inv.t() @ r1inv)                              # re-orthogonalise over all n + k rows
    r2 = torch.linalg.cholesky(g2, upper=True)
    r2inv = torch.linalg.solve_triangular(r2, eye, upper=True)
    q1t = r2inv.t() @ q1t
    rdiag = (r2 @ r1).diagonal()
    logdet = 2.0 * rdiag.abs().log().sum() + (n - k) * torch.log(s2) + extra_logdet
    q1t_pad = torch.zeros(k, ld, device=dev, dtype=torch.float64)              # kept in float64: see Preconditioner.apply_
    q1t_pad[:, :n] = q1t
    lt_pad = torch.zeros(k, ld, device=dev, dtype=wd)
    lt_pad[:, :n] = lt[:, :n]
    return Preconditioner(q1t_pad, s2.to(wd), logdet.to(wd), lt_pad, dinv_sqrt=dinv_sqrt)


def pivoted_cholesky_rows(row_fn, kdiag: torch.Tensor, rank: int, tol: float) -> torch.Tensor:
    """Pivoted Cholesky of a matrix that is only available ROW BY ROW -- the noise-free part of the structured
    operators (sum of kernels, Kronecker K_XX (x) K_TT, Hadamard K_X o K_TT[ti, ti]), whose rows are combinations of rows of their
    members' fused kernels (``gpamd_kernel_rows_f32``).  ``row_fn(p)``: row p ([n]) for a 1-element device index tensor;
    ``kdiag``: the diagonal [n].  Same greedy rule as ``gpamd_pivoted_cholesky_f32`` (largest remaining diagonal, ties by position;
    stop once the remaining trace falls below ``tol`` x the largest diagonal entry).  No host synchronisation: every one of the
    ``rank`` steps is enqueued, steps after the tolerance is met write zero rows (a zero column of L leaves P unchanged).
    Returns L^T as [rank, n] float32 / float64 (the dtype of ``kdiag``)."""
    n = kdiag.numel()
    rank = min(rank, n)
    dev, wd = kdiag.device, kdiag.dtype
    d = kdiag.clone()
    orig = d.max()
    lt = torch.zeros(rank, n, device=dev, dtype=wd)
    used = torch.zeros(n, device=dev, dtype=torch.bool)
    ninf = torch.full((), float("-inf"), device=dev, dtype=wd)
    zero = torch.zeros((), device=dev, dtype=wd)
    for m in range(rank):
        rem = torch.where(used, zero, d)
        go = (rem.sum() / orig > tol) & (rem.max() > 0)
        p = torch.where(used, ninf, d).argmax().reshape(1)
        lmm = d[p].clamp_min(1e-30).sqrt()                      # [1]
        row = row_fn(p).reshape(-1).to(wd)
        l = row - (lt[:m, p].t() @ lt[:m]).reshape(-1) if m else row
        l = l / lmm
        l = torch.where(used, zero, l)
        l[p] = lmm
        l = torch.where(go, l, torch.zeros_like(l))
        lt[m] = l
        d = d - l * l
        used[p] = used[p] | go
    return lt


def build_preconditioner_rows(row_fn, kdiag: torch.Tensor, noise: torch.Tensor, noise_is_vector: bool, rank=None, tol=None, min_size=None):
    """The pivoted-Cholesky preconditioner of a structured operator K + D (``pivoted_cholesky_rows`` + ``preconditioner_from_factor``).
    The reference preconditions every ``AddedDiagLinearOperator`` this way, whatever its first summand (``settings.py:6-31``
    ``max_preconditioner_size``, ``kernels/multitask_kernel.py:46-54``); ``None`` when disabled."""
    n = kdiag.numel()
    rank = settings.max_preconditioner_size.resolve(n) if rank is None else rank
    tol = settings.preconditioner_tolerance.value() if tol is None else tol
    min_size = settings.min_preconditioning_size.value() if min_size is None else min_size
    if rank == 0 or n < min_size or not bool((noise.detach() > 0).all()):
        return None
    lt = pivoted_cholesky_rows(row_fn, kdiag.detach(), rank, tol)
    if not bool(torch.isfinite(lt).all()):
        import warnings

        from .linear_cg import NumericalWarning

        warnings.warn("NaNs encountered in preconditioner computation. Attempting to continue without preconditioning.", NumericalWarning)
        return None
    return preconditioner_from_factor(lt, n, noise, kdiag.dtype, noise_is_vector)


def deterministic_probe_matrix(n: int, t: int, device, dtype=torch.float32, shard=None):
    """``settings.deterministic_probes`` (A.5): ONE (n, t) Gaussian matrix, drawn on first use, stored on the setting
    and re-used by every later evaluation (so line searches / L-BFGS see a deterministic objective).  A user-injected
    matrix (``deterministic_probes.probe_vectors = Z``) of the right length is used as it is; matrices the library draws are kept
    per (n, t), so alternating between models of different size does not redraw.  ``None`` when the flag is off.
    ``shard = (t_total, a, b)`` (probe-column sharding): the stored matrix has ``t_total`` columns -- identical on identically seeded
    ranks, as the single-process run would draw it -- and THIS rank's columns [a, b) are returned, so the estimator keeps
    ``t_total`` distinct probes instead of ``world`` copies of the same ``t_local`` ones."""
    if not settings.deterministic_probes.on():
        return None
    t_all = t if shard is None else shard[0]
    z = settings.deterministic_probes.probe_vectors
    if z is None or z.shape[-2] != n or (shard is not None and z.shape[-1] != t_all):
        key = (n, t_all, str(device), dtype)
        z = settings.deterministic_probes._drawn.get(key)
        if z is None:
            z = torch.randn(n, t_all, device=device, dtype=dtype)
            drawn = settings.deterministic_probes._drawn
            while len(drawn) >= settings.deterministic_probes.max_kept:   # bounded: the oldest matrix goes (20 MB each at n = 5e5, 10 probes)
                drawn.pop(next(iter(drawn)))
            drawn[key] = z
        settings.deterministic_probes.probe_vectors = z
    return z if shard is None else z[:, shard[1] : shard[2]]


def probe_vectors(n: int, t: int, precond: Preconditioner | None, device, generator=None, probes=None, dtype=torch.float32):
    """A.5.  Returns (Zt [t, ld] column-normalised, norms [t]).

    ``probes``: optional user-supplied UN-normalised (n, t) matrix (``deterministic_probes``-style
    injection; lets CPU and GPU runs share Z)."""
    ld = B.round_up(n, 4)
    if probes is None:
        probes = deterministic_probe_matrix(n, t, device, dtype)
    if probes is not None:
        t = probes.shape[-1]
    zt = torch.zeros(t, ld, device=device, dtype=dtype)
    if probes is not None:
        zt[:, :n] = probes.to(device=device, dtype=dtype).t()
    elif precond is None:
        r = torch.randint(0, 2, (t, n), device=device, generator=generator, dtype=torch.int8)
        zt[:, :n] = r.to(dtype) * 2 - 1
    else:
        k = precond.lt.shape[0]
        e1 = torch.randn(t, k, device=device, generator=generator, dtype=dtype)
        e2 = torch.randn(t, n, device=device, generator=generator, dtype=dtype)
        zt[:, :n] = (e1 @ precond.lt.to(dtype))[:, :n] + precond.noise_sqrt(n, dtype) * e2
    norms = B.coldot(zt, zt, n).sqrt()
    zt.div_(norms.unsqueeze(-1))
    return zt, norms


def slq_logdet(t_mats: torch.Tensor, n: int) -> torch.Tensor:
    """``StochasticLQ`` on the mBCG tridiagonals: (n/t) sum_j sum_i evec_j[0,i]^2 log(eval_j[i]).
    Negative eigenvalues -> eigenvalue 1 / eigenvector 0 (``lanczos_tridiag_to_diag``); NaN in T -> NaN."""
    t = t_mats.shape[0]
    if bool(torch.isnan(t_mats).any()):
        return torch.tensor(float("nan"), dtype=torch.float64)
    evals, evecs = torch.linalg.eigh(t_mats.to(torch.float64))
    neg = evals < 0
    evals = evals.masked_fill(neg, 1.0)
    evecs = evecs.masked_fill(neg.unsqueeze(-2), 0.0)
    w = evecs[:, 0, :].pow(2)
    return (w * evals.log()).sum() * (n / t)


@dataclass
class InvQuadLogdetResult:
    inv_quad: torch.Tensor       # [c] per rhs column (device float32)
    logdet: torch.Tensor         # scalar (device float32)
    solves_t: torch.Tensor       # [t + c, ld] probe-major solves (probes first); sharded: the c rhs solves are the owner's, broadcast
    zt: torch.Tensor             # [t, ld] normalised probes
    znorm: torch.Tensor          # [t]
    precond: Preconditioner | None
    info: CGInfo
    logdet_pinvk: torch.Tensor   # SLQ part (before adding log|P|); local partial sum when sharded
    owns_rhs: bool = True        # sharded: this rank solved the rhs columns (their gradient contribution is counted here only)


def inv_quad_logdet_forward(
    x: B.PreparedPoints,
    scale,
    sigma2: torch.Tensor,
    rhs_t: torch.Tensor,
    num_probes=None,
    precond="auto",
    probes=None,
    generator=None,
    tolerance=None,
    max_iter=None,
    group=None,
    t_total=None,
    dvec=None,
    rhs_owner: int = 0,
    kv_partials=None,
    nvec=None,
    row_group=None,
) -> InvQuadLogdetResult:
    """A.6 forward for K_hat = scale*K(x,x) + sigma2*I.

    rhs_t: [c, ld] probe-major ``inv_quad_rhs`` (usually c = 1: y - mu).
    With ``group`` set the probe columns are sharded over ranks: ``num_probes`` / ``probes`` are THIS rank's probes,
    ``t_total`` the global probe count (default: all-reduced sum).  The rhs columns are solved by ONE rank
    (``rhs_owner``, a rank of the group; default 0) -- the others carry probes only -- and their solves / inverse quadratic
    forms are broadcast once at the end.
    STRUCTURED operators (sum of kernels, Kronecker, Hadamard): ``x = None``, ``kv_partials(Dt) -> (P, S, ldp)`` is the noise-free
    product (as for :func:`linear_cg`), ``nvec`` the vector length, ``dvec`` the WHOLE diagonal and ``precond`` an explicit
    :class:`Preconditioner` (``build_preconditioner_rows``) or None; everything else -- probes from N(0, P), sharding, SLQ -- is shared.  Communication: the 2-float stopping-rule all-reduce per CG iteration (stream-
    ordered under RCCL), one scalar all-reduce of the SLQ sums, one broadcast of the c rhs solves.
    TWO-DIMENSIONAL split (``row_group``, with or without ``group``): the ROWS of the system are sharded over ``row_group`` as well
    (:class:`gpytorch_amd.distributed.RowShard`): every rank of a row group carries the same probe columns on its own block of rows, the
    search directions are all-gathered and the solver's inner products all-reduced over the row group each iteration, the stopping rule and
    the SLQ sums go over the probe group as before.  That is what keeps 64 + 1 columns per GPU when 256 probes meet 8 GPUs (4 probe groups x 2
    row halves) instead of 32 + 1, where kernel generation no longer hides under the contraction.  The probes are drawn by the first
    rank of each row group and broadcast (n t floats, once per evaluation); the solves are gathered to full length at the end, so the result
    -- and the backward pass built on it -- looks exactly like the probe-sharded one on every rank."""
    n = x.n if nvec is None else nvec
    dev = rhs_t.device
    wd = x.dtype if x is not None else rhs_t.dtype
    t = settings.num_trace_samples.value() if num_probes is None else num_probes
    if precond == "auto":
        precond = build_preconditioner(x, scale, sigma2, dvec=dvec) if x is not None else None
    if probes is None and settings.deterministic_probes.on():
        shard = None
        if group is not None and t_total is not None:
            from .distributed import probe_shard

            a, b = probe_shard(t_total, torch.distributed.get_world_size(group), torch.distributed.get_rank(group))
            shard = (t_total, a, b)
        probes = deterministic_probe_matrix(n, t, dev, wd, shard=shard)
    if probes is not None:
        t = probes.shape[-1]
    zt, znorm = probe_vectors(n, t, precond, dev, generator, probes, dtype=wd)
    owns_rhs = True
    if group is not None:
        owns_rhs = torch.distributed.get_rank(group) == rhs_owner
    if t_total is None:
        t_total = t
        if group is not None:
            tt = torch.tensor([float(t)], device=dev, dtype=torch.float32)
            allreduce_sum_(tt, group)
            t_total = int(tt.item())
    c = rhs_t.shape[0]
    full = torch.cat([zt, rhs_t.to(wd)], dim=0).contiguous() if owns_rhs else zt
    rs = None
    if row_group is not None and torch.distributed.get_world_size(row_group) > 1:
        if x is None or kv_partials is not None or dvec is not None or not x.fused:
            raise NotImplementedError("row-sharded MLL solves: single fused float32 kernel operator with a constant diagonal")
        from .distributed import RowShard

        rs = RowShard(x, row_group)
        rs.broadcast(zt)                      # one probe draw per row group (the first rank's)
        rs.broadcast(znorm)
        full = torch.cat([zt, rhs_t.to(wd)], dim=0).contiguous() if owns_rhs else zt
        loc, info = linear_cg(None, scale, sigma2, rs.local(full), n_tridiag=t, tolerance=tolerance, max_iter=max_iter, preconditioner=precond,
                              group=group, row_shard=rs)
        solves_t = torch.zeros(loc.shape[0], B.round_up(n, 4), device=dev, dtype=loc.dtype)
        solves_t[:, :n] = rs.gather(loc)[:, :n]
    else:
        solves_t, info = linear_cg(
            x, scale, sigma2, full, n_tridiag=t, tolerance=tolerance, max_iter=max_iter, preconditioner=precond, group=group,
            dvec=dvec, kv_partials=kv_partials, nvec=nvec,
        )
    if settings.skip_logdet_forward.on():
        ld_slq = torch.zeros((), dtype=torch.float64)
    else:
        # (n / t_total) * sum over THIS rank's probes
        ld_slq = slq_logdet(info.t_mats, n) * (t / t_total)
    ld_slq = ld_slq.to(device=dev, dtype=wd)
    if group is not None:
        allreduce_sum_(ld_slq, group)
        if not owns_rhs:
            solves_t = torch.cat([solves_t, torch.zeros(c, solves_t.shape[1], device=dev, dtype=solves_t.dtype)], dim=0)
        ysol = solves_t[t : t + c].contiguous()
        broadcast_(ysol, rhs_owner, group)
        solves_t[t : t + c] = ysol
    logdet = ld_slq + (precond.logdet if precond is not None else 0.0)
    if (settings.rhs_refinement.on() and owns_rhs and x is not None and x.fused and kv_partials is None and rs is None and group is None
            and solves_t.dtype == torch.float32):
        ysol = solves_t[t : t + c].contiguous()
        refine_solves_(x, scale, sigma2, rhs_t.to(wd), ysol, tolerance, max_iter, precond, dvec)
        solves_t[t : t + c] = ysol
    inv_quad = B.coldot(solves_t[t : t + c], rhs_t.to(wd), n)
    return InvQuadLogdetResult(inv_quad, logdet, solves_t, zt, znorm, precond, info, ld_slq, owns_rhs)


def backward_vectors(res: InvQuadLogdetResult, g_iq: torch.Tensor, g_ld: torch.Tensor, t_total: int):
    """The left / right vectors of the A.6 backward, d/dtheta [g_iq . inv_quad + g_ld logdet] = sum_c left[c]^T (dK_hat/dtheta) right[c]:
    left = [K^-1 z_j |z_j| g_ld / t_total  |  -K^-1 y g_iq], right = [P^-1 z_j |z_j|  |  K^-1 y].  Probe-sharded: a rank that does not
    own the rhs columns contributes its probe block only (the owner adds the rhs block once; gradients are all-reduced by the caller).
    Returns (left, right, s_y) with s_y = K^-1 y ([c, ld], every rank has it)."""
    t = res.zt.shape[0]
    wd = res.solves_t.dtype
    c = res.solves_t.shape[0] - t
    g_iq = g_iq.to(wd).reshape(c, 1)
    g_ld = g_ld.to(wd).reshape(())
    s_z = res.solves_t[:t] * res.znorm.unsqueeze(-1)
    s_y = res.solves_t[t:]
    zr = res.zt * res.znorm.unsqueeze(-1)
    if res.precond is not None:
        zr = res.precond.apply_(zr, torch.zeros_like(zr))
    if res.owns_rhs:
        left = torch.cat([s_z * (g_ld / t_total), -s_y * g_iq], dim=0).contiguous()
        right = torch.cat([zr, s_y], dim=0).contiguous()
    else:
        left = (s_z * (g_ld / t_total)).contiguous()
        right = zr.contiguous()
    return left, right, s_y


def allreduce_grads_(grads: list, group):
    """Sum a list of (possibly None) gradient tensors over the probe group in ONE packed all-reduce (in place)."""
    live = [g for g in grads if g is not None]
    if group is None or not live:
        return grads
    wide = torch.float64 if any(g.dtype == torch.float64 for g in live) else torch.float32   # float64 models: no truncation under sharding
    pack = torch.cat([g.reshape(-1).to(wide) for g in live])
    allreduce_sum_(pack, group)
    o = 0
    for g in live:
        g.copy_(pack[o : o + g.numel()].reshape(g.shape).to(g.dtype))
        o += g.numel()
    return grads


def structured_opts(opts: dict, device) -> dict:
    """Solver options of a structured-operator MLL evaluation: ``bbmm_opts`` completed with the ``settings.sharding`` probe group
    (each rank draws its share of ``num_trace_samples`` from a rank-specific generator) -- the same completion
    ``FusedKernelAddedDiagLinearOperator._iql_opts`` performs for the single-kernel operator."""
    group = opts.get("group")
    if group is None:   # (structured operators: probe columns only -- sharding("auto") then means every rank is a probe share)
        group = settings.sharding.mll_groups(0, settings.num_trace_samples.value(), allow_rows=False)[0]
    if group is None or "group" in opts or torch.distributed.get_world_size(group) == 1:
        return opts
    from .distributed import probe_shard

    world, rank = torch.distributed.get_world_size(group), torch.distributed.get_rank(group)
    t_total = settings.num_trace_samples.value()
    a, b = probe_shard(t_total, world, rank)
    if b - a < 1:
        raise ValueError(f"probe sharding needs num_trace_samples >= world size ({t_total} < {world})")
    opts = dict(opts, group=group, num_probes=b - a, t_total=t_total)
    if "generator" not in opts and opts.get("probes") is None:
        opts["generator"] = settings.sharding.rank_generator(group, device)
    return opts


def refine_with_(rhs64: torch.Tensor, sol: torch.Tensor, matvec64, solve_lowp, steps: int = 1) -> int:
    """The refinement loop itself, device-agnostic (the CPU suite drives it with torch operators): ``steps`` times
    r = rhs - A sol in float64 (``matvec64``), d = A^-1 r by the low-precision solver (``solve_lowp`` -> (d, iterations)), sol += d.
    Vectors are rows (probe-major).  Returns the solver iterations spent."""
    extra = 0
    for _ in range(int(steps)):
        r = (rhs64 - matvec64(sol.to(torch.float64))).to(sol.dtype).contiguous()
        delta, its = solve_lowp(r)
        sol += delta
        extra += int(its)
    return extra


def matvec64(x: B.PreparedPoints, scale, sigma2, dvec=None):
    """a64 [c, ld] (probe-major, float64) -> K_hat a in float64: ONE fused float64 product on the same prepared points widened to float64 -- the
    operator the float32 kernels approximate (``csrc/kv_f64.hpp`` for d <= 16).  Many columns go in groups of 80 (the widest tile of that kernel)."""
    x64 = B.PreparedPoints(x.xp.to(torch.float64), x.n, x.d, x.dp, x.kind, x.param)
    sc64 = None if scale is None else scale.detach().to(torch.float64)
    s264 = None if sigma2 is None else sigma2.detach().to(torch.float64)
    dv64 = None if dvec is None else dvec.to(torch.float64)

    def mv(a64):
        if a64.shape[0] <= 80:
            return B.kv(x64, x64, a64, scale=sc64, dscale=s264, vd=a64, dvec=dv64)
        out = torch.empty_like(a64)
        for c0 in range(0, a64.shape[0], 80):
            blk = a64[c0 : c0 + 80].contiguous()
            out[c0 : c0 + 80] = B.kv(x64, x64, blk, scale=sc64, dscale=s264, vd=blk, dvec=dv64)
        return out

    return mv


def refine_solves_(x: B.PreparedPoints, scale, sigma2, rhs_t, sol_t, tolerance=None, max_iter=None, precond=None, dvec=None, steps=None):
    """``settings.rhs_refinement``: in-place mixed-precision iterative refinement of float32 solves ``sol_t`` ([c, ld], probe-major) of
    K_hat X = rhs.  Per step: r = rhs - K_hat sol with ONE fused float64 product on the same prepared points (widened to float64: the operator
    the float32 kernels approximate), then K_hat d = r by float32 mBCG, sol += d.  Returns the extra CG iterations."""
    steps = settings.rhs_refinement.steps if steps is None else steps

    def solve32(r):
        delta, info = linear_cg(x, scale, sigma2, r, n_tridiag=0, tolerance=tolerance, max_iter=max_iter, preconditioner=precond, dvec=dvec)
        return delta, info.iterations

    return refine_with_(rhs_t.to(torch.float64), sol_t, matvec64(x, scale, sigma2, dvec), solve32, steps)


def variational_inv_quad(matmul64, b: torch.Tensor, x: torch.Tensor, cols: int = 80, rows: int = 32768) -> torch.Tensor:
    """B^T A^-1 B ([m, m], float64) from an APPROXIMATE float32 solve X ~ A^-1 B, to SECOND order in its error, with one float64 product and
    NO second solve:  B^T A^-1 B = X^T (2 B - A X) + E^T A E  for X = A^-1 B - E, where ``matmul64`` ([n, k] float64 -> A @ it, float64)
    supplies A X.  The remainder E^T A E is the squared ENERGY norm of the solve error -- the very quantity CG minimises -- so a float32 mBCG
    solve at relative accuracy 3e-4 leaves ~1e-7 of the quadratic form, where the plain contraction B^T X keeps the first-order 3e-4: what the
    predictive variance of f = K_** - K_*X K_hat^-1 K_X* (1 - 0.9998.. at a well-determined point) needs, at 1.2-1.3 x the time of the
    unrefined solve instead of the 3.7 x of a refined one (two solves + the same float64 product).  ``b`` / ``x``: [n, m] (any float dtype);
    work proceeds in groups of ``cols`` columns and ``rows`` rows so that no float64 copy of the n x m operands is ever held."""
    n, m = b.shape
    out = torch.empty(m, m, device=b.device, dtype=torch.float64)
    for c0 in range(0, m, cols):
        xg = x[:, c0 : c0 + cols].to(torch.float64)
        y = 2.0 * b[:, c0 : c0 + cols].to(torch.float64) - matmul64(xg)          # [n, k]: 2 B_g - A X_g
        acc = torch.zeros(m, y.shape[1], device=b.device, dtype=torch.float64)
        for r0 in range(0, n, rows):
            acc.addmm_(x[r0 : r0 + rows].to(torch.float64).t(), y[r0 : r0 + rows])
        out[:, c0 : c0 + cols] = acc
    return 0.5 * (out + out.t())


def solve(x: B.PreparedPoints, scale, sigma2, rhs_t, tolerance=None, max_iter=None, precond="auto"):
    """K_hat^-1 rhs by preconditioned mBCG (``LinearOperator.solve`` on the CG branch, A.1)."""
    if precond == "auto":
        precond = build_preconditioner(x, scale, sigma2)
    sol, info = linear_cg(x, scale, sigma2, rhs_t, n_tridiag=0, tolerance=tolerance, max_iter=max_iter, preconditioner=precond)
    if settings.rhs_refinement.on() and x is not None and x.fused and sol.dtype == torch.float32:
        refine_solves_(x, scale, sigma2, rhs_t, sol, tolerance, max_iter, precond)
    return sol, info


LOG_2PI = math.log(2 * math.pi)
