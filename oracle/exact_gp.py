"""Oracle: ExactGP marginal log-likelihood and predictive posterior.  Test infrastructure only.

Two restatements of the same quantities:

* ``dense_*``  -- dense float64 Cholesky: the deterministic ground truth every reference test
  compares against (``test/lazy/test_lazy_evaluated_kernel_tensor.py:88-92``,
  ``test/distributions/test_multivariate_normal.py:226-228``,
  ``test/examples/test_simple_gp_regression.py:386-388``).
* ``bbmm_*``   -- the reference's fast path as actually executed with ``max_cholesky_size(0)``:
  ``likelihoods/gaussian_likelihood.py:117-121`` (K + s2 I) ->
  ``distributions/multivariate_normal.py:221-252`` (log_prob via inv_quad_logdet) ->
  ``mlls/exact_marginal_log_likelihood.py:83-89`` (divide by n); prediction per
  ``models/exact_prediction_strategies.py:278-286`` (mean cache by CG, ``exact_gp.py:324`` eval
  tolerance 0.01), ``:371-412`` (mean), ``:414-478`` (covariance: solve path / LOVE path).
  The linear_operator pieces (mBCG, SLQ, pivoted Cholesky, Lanczos) are the restatements in this
  package (iteration-level parity UNPINNED, see package docstring).
"""
from __future__ import annotations

import math

import torch

from . import kernels as K_
from .lanczos import root_inv_decomposition
from .linear_cg import linear_cg
from .pivoted_cholesky import build_preconditioner, pivoted_cholesky, probe_vectors
from .slq import inv_quad_logdet


def _khat(kind, X, ls, os_, s2):
    n = X.shape[-2]
    return K_.kernel_matrix(kind, X, X, ls, os_, x1_eq_x2=True) + s2 * torch.eye(n, dtype=X.dtype)


def dense_log_prob(Khat: torch.Tensor, diff: torch.Tensor) -> torch.Tensor:
    """MVN log-prob by Cholesky (multivariate_normal.py:251: -0.5*(inv_quad + logdet + n log 2pi))."""
    n = diff.shape[-1]
    Lc = torch.linalg.cholesky(Khat)
    sol = torch.cholesky_solve(diff.unsqueeze(-1), Lc).squeeze(-1)
    inv_quad = (diff * sol).sum(-1)
    logdet = 2.0 * Lc.diagonal(dim1=-2, dim2=-1).log().sum(-1)
    return -0.5 * (inv_quad + logdet + n * math.log(2 * math.pi))


def dense_mll(kind, X, y, ls, os_, s2, mean=0.0):
    """ExactMarginalLogLikelihood value (exact_marginal_log_likelihood.py:83-89), no priors."""
    return dense_log_prob(_khat(kind, X, ls, os_, s2), y - mean) / y.shape[-1]


def dense_mll_and_grads(kind, X, y, ls, os_, s2, mean=0.0):
    """Value and d/d(lengthscale, outputscale, noise) by float64 autograd through Cholesky."""
    p = [torch.as_tensor(v, dtype=X.dtype).clone().requires_grad_(True) for v in (ls, os_, s2)]
    val = dense_mll(kind, X, y, p[0], p[1], p[2], mean)
    g = torch.autograd.grad(val, p)
    return val.detach(), [gi.detach() for gi in g]


def dense_solve_logdet(kind, X, rhs, ls, os_, s2):
    Kh = _khat(kind, X, ls, os_, s2)
    Lc = torch.linalg.cholesky(Kh)
    return torch.cholesky_solve(rhs, Lc), 2.0 * Lc.diagonal().log().sum()


def dense_posterior(kind, X, y, Xs, ls, os_, s2, mean=0.0, noise=True):
    """Predictive mean / variance at Xs (exact_prediction_strategies.py:371-478 + likelihood noise)."""
    Kh = _khat(kind, X, ls, os_, s2)
    Lc = torch.linalg.cholesky(Kh)
    # the joint kernel call centres Matern inputs with the mean of cat(train, test): stationary, so
    # any common shift gives identical values in exact arithmetic
    Ksx = K_.kernel_matrix(kind, Xs, X, ls, os_, x1_eq_x2=False)
    alpha = torch.cholesky_solve((y - mean).unsqueeze(-1), Lc)
    mu = (Ksx @ alpha).squeeze(-1) + mean
    v = torch.linalg.solve_triangular(Lc, Ksx.t(), upper=False)
    var = os_ * torch.ones(Xs.shape[-2], dtype=X.dtype) - v.pow(2).sum(0)
    if noise:
        var = var + s2
    return mu, var


# ----------------------------------------------------------------------------------------------
# BBMM path (restated): what gpytorch + linear_operator execute with max_cholesky_size(0)
# ----------------------------------------------------------------------------------------------

def make_matmul(kind, X, ls, os_, s2, dense=True, chunk=4096):
    """K_hat @ V closure: dense (default gpytorch behaviour: materialise K once, matmul per
    iteration) or chunked matrix-free (lazy_evaluated_kernel_tensor.py:245-275)."""
    if dense:
        Kmat = K_.kernel_matrix(kind, X, X, ls, os_, x1_eq_x2=True)

        def mm(V):
            return Kmat @ V + s2 * V
    else:
        def mm(V):
            return K_.kernel_matmul_chunked(kind, X, X, ls, os_, V, chunk) + s2 * V
    return mm


def make_preconditioner(kind, X, ls, os_, s2, rank, error_tol=1e-3, min_size=2000):
    """A.4 gating + construction.  Returns (apply|None, logdet_P, L|None)."""
    n = X.shape[-2]
    if rank == 0 or n < min_size:
        return None, 0.0, None
    diag = torch.full((n,), float(os_), dtype=X.dtype)

    def row_fn(p):
        return K_.kernel_matrix(kind, X[p : p + 1], X, ls, os_, x1_eq_x2=False, direct=True).reshape(-1)

    L = pivoted_cholesky(diag, row_fn, rank, error_tol)
    apply, logdet, _ = build_preconditioner(L, float(s2))
    return apply, logdet, L


def bbmm_mll(
    kind, X, y, ls, os_, s2, mean=0.0, num_probes=10, precond_rank=15, min_precond_size=2000,
    cg_tol=1.0, max_cg_iter=1000, max_lanczos_iter=20, probes=None, seed=1234, dense=True, return_aux=False,
    precond_L=None,
):
    """MLL through mBCG + SLQ.  ``probes``: optional pre-drawn (n, t) UN-normalised probe matrix
    (so GPU and CPU runs can share Z); otherwise drawn per A.5 with ``seed``.
    ``precond_L``: optional (n, k) pivoted-Cholesky factor to build the preconditioner from (test aid:
    float32 ties make the pivot sequence implementation-dependent, and the SLQ / trace estimators
    are only unbiased when the probes' covariance is the preconditioner actually applied)."""
    n = X.shape[-2]
    mm = make_matmul(kind, X, ls, os_, s2, dense)
    if precond_L is not None:
        L = precond_L.to(X.dtype)
        papply, plogdet, _ = build_preconditioner(L, float(s2))
    else:
        papply, plogdet, L = make_preconditioner(kind, X, ls, os_, s2, precond_rank, min_size=min_precond_size)
    if probes is None:
        g = torch.Generator().manual_seed(seed)
        Z, Znorm = probe_vectors(n, num_probes, L, float(s2), g, X.dtype)
    else:
        Znorm = probes.norm(2, dim=-2, keepdim=True)
        Z = probes / Znorm
    diff = (y - mean).unsqueeze(-1)
    iq, ld, aux = inv_quad_logdet(
        mm, n, diff, Z, papply, plogdet, tolerance=cg_tol, max_iter=max_cg_iter,
        max_tridiag_iter=max_lanczos_iter, return_aux=True,
    )
    res = -0.5 * (iq.sum() + ld + n * math.log(2 * math.pi)) / n
    if return_aux:
        aux.update(inv_quad=iq.sum(), logdet=ld, Z=Z, Znorm=Znorm, precond=papply, L=L)
        return res, aux
    return res


def bbmm_mll_grads(kind, X, aux, ls, os_, s2, g_out=1.0):
    """A.6 backward: gradient of the MLL wrt (lengthscale, outputscale, noise) given the forward's
    solves.  The bilinear derivative sum_c left[:,c]^T dK_hat right[:,c] is taken by float64
    autograd through the dense kernel (what DenseLinearOperator._bilinear_derivative does)."""
    n = X.shape[-2]
    solves, Z, Znorm, papply = aux["solves"], aux["Z"], aux["Znorm"], aux["precond"]
    t = Z.shape[-1]
    g_iq = g_ld = -0.5 / n * g_out
    S_z = solves[:, :t] * Znorm
    S_y = solves[:, t:]
    Zr = Z * Znorm
    if papply is not None:
        Zr = papply(Zr)
    left = torch.cat([S_z * (g_ld / t), -S_y * g_iq], dim=-1)
    right = torch.cat([Zr, S_y], dim=-1)
    p = [torch.as_tensor(v, dtype=X.dtype).clone().requires_grad_(True) for v in (ls, os_, s2)]
    Kh = _khat(kind, X, p[0], p[1], p[2])
    val = (left * (Kh @ right)).sum()
    return [gi.detach() for gi in torch.autograd.grad(val, p)]


def bbmm_posterior(
    kind, X, y, Xs, ls, os_, s2, mean=0.0, eval_cg_tol=0.01, max_cg_iter=1000, precond_rank=15,
    min_precond_size=2000, love_rank=100, fast_pred_var=True, init_vec=None, noise=True, dense=True,
):
    """Predictive mean (CG mean cache) and variance (LOVE root-inverse, or per-test solves)."""
    n = X.shape[-2]
    mm = make_matmul(kind, X, ls, os_, s2, dense)
    papply, _, _ = make_preconditioner(kind, X, ls, os_, s2, precond_rank, min_size=min_precond_size)
    mean_cache = linear_cg(mm, (y - mean).unsqueeze(-1), tolerance=eval_cg_tol, max_iter=max_cg_iter, preconditioner=papply)
    Ksx = K_.kernel_matrix(kind, Xs, X, ls, os_, x1_eq_x2=False)
    mu = (Ksx @ mean_cache).squeeze(-1) + mean
    prior_var = os_ * torch.ones(Xs.shape[-2], dtype=X.dtype)
    if fast_pred_var:
        if init_vec is None:
            init_vec = torch.randn(n, 1, generator=torch.Generator().manual_seed(7), dtype=X.dtype)
        Rinv = root_inv_decomposition(mm, n, love_rank, init_vec)
        var = prior_var - (Ksx @ Rinv).pow(2).sum(-1)
    else:
        sol = linear_cg(mm, Ksx.t().contiguous(), tolerance=eval_cg_tol, max_iter=max_cg_iter, preconditioner=papply)
        var = prior_var - (Ksx * sol.t()).sum(-1)
    if noise:
        var = var + s2
    return mu, var
