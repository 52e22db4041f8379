"""Dense float64 ground truth ON THE DEVICE for the at-size MLL checks (test infrastructure, never imported by the product).

The reference's own parity tests all compare with a dense factorisation (``test/lazy/test_lazy_evaluated_kernel_tensor.py:84-105``,
``test/distributions/test_multivariate_normal.py:219-237``).  An MI355X holds a 1e5 x 1e5 float64 matrix (80 GB of 288 GB), so the
same comparison can be made at BASELINE's C2 size: K_hat is built in float64, row block by row block, with plain torch and the
reference's dense formulas

  * ``gpytorch/kernels/kernel.py:26-49``            sq_dist: mean-centred Gram trick, diagonal forced to 0, clamp >= 0
  * ``gpytorch/kernels/kernel.py:52-60``            dist = sqrt(clamp(sq_dist, 1e-30))
  * ``gpytorch/functions/rbf_covariance.py:14-19``  exp(-sq_dist(x / l) / 2)
  * ``gpytorch/functions/matern_covariance.py:18-50`` (inputs centred by x1.mean(-2) first)
  * ``gpytorch/kernels/scale_kernel.py:108-118``, ``gpytorch/likelihoods/gaussian_likelihood.py:117-121``  theta K + sigma^2 I

and factorised IN PLACE by a blocked right-looking Cholesky (diagonal block: torch.linalg.cholesky, panel: triangular solve,
trailing update: float64 GEMMs on the lower block triangle only), so that the peak footprint is the matrix itself plus one panel.
"""
from __future__ import annotations

import math

import torch


def dense_khat(kind: str, X: torch.Tensor, lengthscale, outputscale: float, sigma2: float, device, block: int = 4096) -> torch.Tensor:
    """theta * K(X, X) + sigma2 * I as an n x n float64 device tensor (X: [n, d] on any device; lengthscale scalar or [d])."""
    n = X.shape[0]
    x = X.to(device=device, dtype=torch.float64)
    ls = torch.as_tensor(lengthscale, dtype=torch.float64, device=device).reshape(1, -1)
    if kind != "rbf":
        x = x - x.mean(-2, keepdim=True)              # matern_covariance.py:19-21 (centre, then divide)
    x = x / ls
    x1 = x - x.mean(-2, keepdim=True)                  # sq_dist's adjustment (kernel.py:29-30)
    norm = x1.pow(2).sum(-1, keepdim=True)
    one = torch.ones_like(norm)
    a_ = torch.cat([-2.0 * x1, norm, one], -1)         # kernel.py:41-42
    b_ = torch.cat([x1, one, norm], -1).t().contiguous()
    K = torch.empty(n, n, device=device, dtype=torch.float64)
    nu = {"matern12": 0.5, "matern32": 1.5, "matern52": 2.5}.get(kind)
    for a in range(0, n, block):
        e = min(n, a + block)
        res = a_[a:e] @ b_
        idx = torch.arange(a, e, device=device)
        res[idx - a, idx] = 0.0                        # kernel.py:45-46: x1_eq_x2 -> exact zeros on the diagonal
        res.clamp_min_(0.0)
        if kind == "rbf":
            res.mul_(-0.5).exp_()
        else:
            s = res.clamp_min_(1e-30).sqrt_().mul_(math.sqrt(2 * nu))
            ex = torch.exp(-s)
            if nu == 0.5:
                res = ex
            elif nu == 1.5:
                res = s.add_(1.0).mul_(ex)
            else:
                res = (s.pow(2) / 3).add_(s).add_(1.0).mul_(ex)
            del ex
        if outputscale != 1.0:
            res.mul_(outputscale)
        res[idx - a, idx] += sigma2
        K[a:e] = res
        del res
    return K


def cholesky_inplace_(K: torch.Tensor, block: int = 4096) -> torch.Tensor:
    """Blocked right-looking Cholesky on the LOWER triangle of K, in place (the strict upper triangle is left untouched
    and must not be read afterwards).  Returns K."""
    n = K.shape[0]
    for j in range(0, n, block):
        e = min(n, j + block)
        ljj = torch.linalg.cholesky(K[j:e, j:e])
        K[j:e, j:e] = ljj
        if e == n:
            break
        # panel: P = K[e:, j:e] L_jj^-T
        P = torch.linalg.solve_triangular(ljj, K[e:, j:e].t(), upper=False).t().contiguous()
        K[e:, j:e] = P
        # trailing update, lower block triangle only: K[i, e..i] -= P_i P_{e..i}^T
        for i in range(e, n, block):
            ie = min(n, i + block)
            K[i:ie, e:ie] -= P[i - e : ie - e] @ P[: ie - e].t()
        del P
    return K


def logdet_from_factor(L: torch.Tensor) -> float:
    return float(2.0 * L.diagonal().log().sum())


def solve_with_factor(L: torch.Tensor, b: torch.Tensor, block: int = 4096) -> torch.Tensor:
    """K^-1 b from the in-place factor (only the lower triangle of L is read).  b: [n, c] float64."""
    n = L.shape[0]
    y = b.clone()
    for j in range(0, n, block):               # forward: L y = b
        e = min(n, j + block)
        if j:
            y[j:e] -= L[j:e, :j] @ y[:j]
        y[j:e] = torch.linalg.solve_triangular(torch.tril(L[j:e, j:e]), y[j:e], upper=False)
    starts = list(range(0, n, block))
    for j in reversed(starts):                  # backward: L^T x = y
        e = min(n, j + block)
        if e < n:
            y[j:e] -= L[e:, j:e].t() @ y[e:]
        y[j:e] = torch.linalg.solve_triangular(torch.tril(L[j:e, j:e]).t(), y[j:e], upper=True)
    return y


def dense_truth(kind, X, y, lengthscale, outputscale, sigma2, device, block=4096):
    """(inv_quad = y^T K_hat^-1 y, logdet K_hat, K_hat^-1 y) in float64 from the dense factorisation."""
    K = dense_khat(kind, X, lengthscale, outputscale, sigma2, device, block)
    cholesky_inplace_(K, block)
    ld = logdet_from_factor(K)
    yd = y.to(device=device, dtype=torch.float64).reshape(-1, 1)
    sol = solve_with_factor(K, yd, block)
    iq = float((sol * yd).sum())
    del K
    return iq, ld, sol.squeeze(-1)
