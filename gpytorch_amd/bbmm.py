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


def build_preconditioner(x: B.PreparedPoints, scale, sigma2: torch.Tensor, rank=None, tol=None, min_size=None):
    """Pivoted-Cholesky preconditioner P = L L^T + s2 I (constant-diagonal branch of A.4).

    Returns ``None`` when disabled (rank 0 or n < min_preconditioning_size), else a
    :class:`Preconditioner` holding Q1 (as [k, ld] rows), log|P| and L^T.

    The reference takes a Householder QR of [L; sqrt(s2) I]; here the same thin factor is obtained
    with two rounds of Cholesky-QR (GEMM + k x k Cholesky in float64 + triangular solve), which
    keeps the n x k work on rocBLAS GEMMs.  Q1 Q1^T and |diag R| -- the only quantities used -- are
    identical up to rounding."""
    rank = settings.max_preconditioner_size.value() if rank is None else rank
    tol = settings.preconditioner_tolerance.value() if tol is None else tol
    min_size = settings.min_preconditioning_size.value() if min_size is None else min_size
    n = x.n
    if rank == 0 or n < min_size or float(sigma2.detach().reshape(-1)[0]) <= 0.0:
        return None  # (P = L L^T + s2 I is singular without a positive diagonal)
    lt, _, k = B.pivoted_cholesky(x, scale, rank, tol)  # [k, n]
    if not bool(torch.isfinite(lt).all()):
        import warnings

        from .linear_cg import NumericalWarning

        warnings.warn("NaNs encountered in preconditioner computation. Attempting to continue without preconditioning.", NumericalWarning)
        return None
    dev = lt.device
    wd = x.dtype
    ld = B.round_up(n, 4)
    s2 = sigma2.detach().reshape(()).to(torch.float64)
    eye = torch.eye(k, device=dev, dtype=torch.float64)
    # Cholesky-QR twice, all in float64 (n x k GEMMs: 2 n k^2 flop each, < 10 ms at n = 5e5, k = 100)
    ltd = lt.to(torch.float64)
    r1 = torch.linalg.cholesky(ltd @ ltd.t() + s2 * eye, upper=True)           # G = L^T L + s2 I = R1^T R1
    r1inv = torch.linalg.solve_triangular(r1, eye, upper=True)
    q1t = r1inv.t() @ ltd                                                       # [k, n] = (L R1^-1)^T
    g2 = q1t @ q1t.t() + s2 * (r1inv.t() @ r1inv)                               # re-orthogonalise over all n + k rows
    r2 = torch.linalg.cholesky(g2, upper=True)
    r2inv = torch.linalg.solve_triangular(r2, eye, upper=True)
    q1t = r2inv.t() @ q1t
    rdiag = (r2 @ r1).diagonal()
    logdet = 2.0 * rdiag.abs().log().sum() + (n - k) * torch.log(s2)
    q1t_pad = torch.zeros(k, ld, device=dev, dtype=torch.float64)              # kept in float64: see Preconditioner.apply_
    q1t_pad[:, :n] = q1t
    lt_pad = torch.zeros(k, ld, device=dev, dtype=wd)
    lt_pad[:, :n] = lt
    return Preconditioner(q1t_pad, sigma2.detach().reshape(()).to(wd), logdet.to(wd), lt_pad)


def deterministic_probe_matrix(n: int, t: int, device, dtype=torch.float32):
    """``settings.deterministic_probes`` (A.5): ONE (n, t) Gaussian matrix, drawn on first use, stored on the setting
    and re-used by every later evaluation (so line searches / L-BFGS see a deterministic objective).  A user-injected
    matrix is used as it is; a stored matrix of the wrong length (another model) is redrawn.  ``None`` when the flag is off."""
    if not settings.deterministic_probes.on():
        return None
    z = settings.deterministic_probes.probe_vectors
    if z is None or z.shape[-2] != n:
        z = torch.randn(n, t, device=device, dtype=dtype)
        settings.deterministic_probes.probe_vectors = z
    return z


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
        zt[:, :n] = (e1 @ precond.lt.to(dtype))[:, :n] + precond.sigma2.to(dtype).sqrt() * e2
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
) -> InvQuadLogdetResult:
    """A.6 forward for K_hat = scale*K(x,x) + sigma2*I.

    rhs_t: [c, ld] probe-major ``inv_quad_rhs`` (usually c = 1: y - mu).
    With ``group`` set the probe columns are sharded over ranks: ``num_probes`` / ``probes`` are THIS rank's probes,
    ``t_total`` the global probe count (default: all-reduced sum).  The rhs columns are solved by ONE rank
    (``rhs_owner``, a rank of the group; default 0) -- the others carry probes only -- and their solves / inverse quadratic
    forms are broadcast once at the end.  Communication: the 2-float stopping-rule all-reduce per CG iteration (stream-
    ordered under RCCL), one scalar all-reduce of the SLQ sums, one broadcast of the c rhs solves."""
    n = x.n
    dev = rhs_t.device
    wd = x.dtype
    t = settings.num_trace_samples.value() if num_probes is None else num_probes
    if dvec is not None:
        precond = None  # the pivoted-Cholesky preconditioner is built for the constant-diagonal branch only (A.4)
    if precond == "auto":
        precond = build_preconditioner(x, scale, sigma2)
    if probes is None:
        probes = deterministic_probe_matrix(n, t, dev, wd)
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
    solves_t, info = linear_cg(
        x, scale, sigma2, full, n_tridiag=t, tolerance=tolerance, max_iter=max_iter, preconditioner=precond, group=group,
        dvec=dvec,
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
    inv_quad = B.coldot(solves_t[t : t + c], rhs_t.to(wd), n)
    return InvQuadLogdetResult(inv_quad, logdet, solves_t, zt, znorm, precond, info, ld_slq, owns_rhs)


def solve(x: B.PreparedPoints, scale, sigma2, rhs_t, tolerance=None, max_iter=None, precond="auto"):
    """K_hat^-1 rhs by preconditioned mBCG (``LinearOperator.solve`` on the CG branch, A.1)."""
    if precond == "auto":
        precond = build_preconditioner(x, scale, sigma2)
    return linear_cg(x, scale, sigma2, rhs_t, n_tridiag=0, tolerance=tolerance, max_iter=max_iter, preconditioner=precond)


LOG_2PI = math.log(2 * math.pi)
