"""Oracle: stochastic Lanczos quadrature log-det and the BBMM ``inv_quad_logdet`` (forward and
hyper-parameter gradient).  Test infrastructure only.

Restates ``linear_operator.functions._inv_quad_logdet.InvQuadLogdet`` +
``utils/stochastic_lq.py`` + ``lanczos_tridiag_to_diag`` (linear_operator v0.6.x, third-party, not
vendored).  Consumer in the reference: ``gpytorch/distributions/
multivariate_normal.py:249-251``.  UNPINNED at iteration level; pinned by comparing, for FIXED
probe vectors, against dense slogdet / Cholesky within the reference's own tolerances.
"""
from __future__ import annotations

import torch

from .linear_cg import linear_cg


def slq_logdet(T: torch.Tensor, n: int) -> torch.Tensor:
    """T: (t, m, m) tridiagonals from mBCG.  (n/t) * sum_j sum_i evec_j[0,i]^2 log(eval_j[i])."""
    t = T.shape[0]
    if torch.isnan(T).any():
        return torch.tensor(float("nan"), dtype=T.dtype)
    evals, evecs = torch.linalg.eigh(T.to(torch.float64))
    neg = evals < 0
    evals = evals.masked_fill(neg, 1.0)
    evecs = evecs.masked_fill(neg.unsqueeze(-2), 0.0)
    w = evecs[:, 0, :].pow(2)
    return ((w * evals.log()).sum() * (n / t)).to(T.dtype)


def inv_quad_logdet(
    matmul_closure,
    n: int,
    inv_quad_rhs: torch.Tensor,
    probes: torch.Tensor,
    precond=None,
    logdet_precond: float | torch.Tensor = 0.0,
    tolerance: float = 1.0,
    max_iter: int = 1000,
    max_tridiag_iter: int = 20,
    return_aux: bool = False,
):
    """A.6 forward.  probes: (n, t) already column-normalised.  rhs = [probes | inv_quad_rhs]."""
    t = probes.shape[-1]
    rhs = torch.cat([probes, inv_quad_rhs], dim=-1)
    solves, T, info = linear_cg(
        matmul_closure, rhs, n_tridiag=t, tolerance=tolerance, max_iter=max_iter,
        max_tridiag_iter=max_tridiag_iter, preconditioner=precond, return_info=True,
    )
    logdet = slq_logdet(T, n) + logdet_precond
    inv_quad = (solves[:, t:] * inv_quad_rhs).sum(-2)
    if return_aux:
        return inv_quad, logdet, dict(solves=solves, T=T, info=info)
    return inv_quad, logdet
