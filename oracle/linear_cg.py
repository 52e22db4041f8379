"""Oracle: modified batched conjugate gradients (mBCG).  Test infrastructure only.

Restates the published algorithm of ``linear_operator.utils.linear_cg`` (linear_operator
v0.6.x, third-party, not vendored under /root/reference).
Reference call sites: ``gpytorch/distributions/multivariate_normal.py:249`` (through
``inv_quad_logdet``) and ``gpytorch/models/exact_prediction_strategies.py:286,444`` (through
``solve``); asserted-called at ``test/lazy/test_lazy_evaluated_kernel_tensor.py:82-111``.

Iteration-level parity with the reference is UNPINNED (no golden vectors exist, the package is absent); results are
pinned against dense Cholesky in tests/test_oracle_bbmm.py, and the recurrences themselves against independent third-party
code in tests/test_oracle_independent_cpu.py (k-step iterates == scipy.sparse.linalg.cg's, with and without a preconditioner;
the alpha / beta tridiagonal == an independent Lanczos tridiagonalisation).

Layout here is the reference's: rhs is (n, c) with one right-hand side per COLUMN.

``mean_residual_fn`` (not in the reference): hook that replaces ``rnorm.mean()`` in the stopping rule,
used by the world_size-2 tests to restate the probe-sharded solve (global mean via all-reduce).
"""
from __future__ import annotations

import warnings

import torch


def linear_cg(
    matmul_closure,
    rhs: torch.Tensor,
    n_tridiag: int = 0,
    tolerance: float = 1.0,
    eps: float = 1e-10,
    stop_updating_after: float = 1e-10,
    max_iter: int = 1000,
    max_tridiag_iter: int = 20,
    initial_guess: torch.Tensor | None = None,
    preconditioner=None,
    return_info: bool = False,
    mean_residual_fn=None,
    rowsum_fn=None,
):
    # rowsum_fn (test aid for ROW-sharded solves): the vectors hold only this rank's rows and every
    # reduction over rows -- inner products, norms -- goes through rowsum_fn (local sum + all-reduce).  None: plain sums.
    if rowsum_fn is None:
        def dot(a_, b_):
            return (a_ * b_).sum(-2, keepdim=True)

        def norm(a_):
            return a_.norm(2, dim=-2, keepdim=True)
    else:
        def dot(a_, b_):
            return rowsum_fn((a_ * b_).sum(-2, keepdim=True))

        def norm(a_):
            return rowsum_fn((a_ * a_).sum(-2, keepdim=True)).sqrt()
    n, c = rhs.shape[-2], rhs.shape[-1]
    if preconditioner is None:
        def preconditioner(x):  # noqa: E306
            return x.clone()

    n_iter = max_iter
    n_tri_iter = min(max_tridiag_iter, n)

    bnorm = norm(rhs)
    zero_rhs = bnorm.lt(eps)
    bnorm = bnorm.masked_fill(zero_rhs, 1)
    B = rhs / bnorm

    X = torch.zeros_like(B) if initial_guess is None else initial_guess.clone()
    R = B - matmul_closure(X)
    if not torch.equal(R, R):
        raise RuntimeError("NaNs encountered when trying to perform matrix-vector multiplication")

    rnorm = norm(R)
    converged = rnorm.lt(stop_updating_after)

    Z = preconditioner(R)
    D = Z.clone()
    rho = dot(R, Z)

    T = torch.zeros(n_tri_iter, n_tri_iter, n_tridiag, dtype=rhs.dtype) if n_tridiag else None
    alpha_hist, beta_hist = [], []
    update_tridiag = True
    last_tridiag_iter = 0
    ainv_prev = b_prev = None
    tolerance_reached = False
    k_done = 0
    min_iter = min(10, max_iter - 1)

    for k in range(n_iter):
        Q = matmul_closure(D)
        den = dot(D, Q)
        bad = den.lt(eps)
        den = den.masked_fill(bad, 1)
        alpha = rho / den
        alpha = alpha.masked_fill(bad, 0)
        alpha = alpha.masked_fill(converged, 0)

        R = R - alpha * Q
        Z = preconditioner(R)
        X = X + alpha * D

        rho_old = rho
        rho = dot(R, Z)
        bad = rho_old.lt(eps)
        rho_old = rho_old.masked_fill(bad, 1)
        beta = rho / rho_old
        beta = beta.masked_fill(bad, 0)
        D = Z + beta * D

        rnorm = norm(R)
        rnorm = rnorm.masked_fill(zero_rhs, 0)
        converged = rnorm.lt(stop_updating_after)
        alpha_hist.append(alpha.reshape(-1).clone())
        beta_hist.append(beta.reshape(-1).clone())
        k_done = k + 1

        if (
            k >= min_iter
            and bool((rnorm.mean() if mean_residual_fn is None else mean_residual_fn(rnorm)) < tolerance)
            and not (n_tridiag and k < min(n_tri_iter, max_iter - 1))
        ):
            tolerance_reached = True
            break

        if n_tridiag and k < n_tri_iter and update_tridiag:
            a = alpha.reshape(-1)[:n_tridiag]
            b = beta.reshape(-1)[:n_tridiag]
            a0 = a.eq(0)
            ainv = 1.0 / torch.where(a0, torch.ones_like(a), a)  # reciprocal is 1 where alpha == 0
            if k == 0:
                T[k, k] = ainv
            else:
                T[k, k] = ainv + b_prev * ainv_prev
                off = b_prev.sqrt() * ainv_prev
                T[k, k - 1] = off
                T[k - 1, k] = off
                if T[k - 1, k].max() < 1e-6:
                    update_tridiag = False
            last_tridiag_iter = k
            ainv_prev, b_prev = ainv, b.clone()

    X = X * bnorm
    if not tolerance_reached and n_iter > 0:
        warnings.warn(
            f"CG terminated in {k_done} iterations with average residual norm {float(rnorm.mean())}"
            f" which is larger than the tolerance of {tolerance}.",
            RuntimeWarning,
        )
    info = dict(iters=k_done, rnorm=rnorm.reshape(-1), alpha=alpha_hist, beta=beta_hist, tolerance_reached=tolerance_reached)
    if n_tridiag:
        m = last_tridiag_iter + 1
        Tm = T[:m, :m].permute(2, 0, 1).contiguous()  # (n_tridiag, m, m)
        return (X, Tm, info) if return_info else (X, Tm)
    return (X, info) if return_info else X
