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


# ---- round 4: derivatives, bilinear forms and the predictive posterior from the same dense float64 machinery -------------------------

def _scaled_operands(kind, X, lengthscale, device):
    x = X.to(device=device, dtype=torch.float64)
    ls = torch.as_tensor(lengthscale, dtype=torch.float64, device=device).reshape(1, -1)
    if kind != "rbf":
        x = x - x.mean(-2, keepdim=True)
    x = x / ls
    x1 = x - x.mean(-2, keepdim=True)
    norm = x1.pow(2).sum(-1, keepdim=True)
    one = torch.ones_like(norm)
    return torch.cat([-2.0 * x1, norm, one], -1), torch.cat([x1, one, norm], -1).t().contiguous()


def kernel_rows_and_dl(kind, a_rows, b_, lengthscale: float, row0=None):
    """(K, dK/dl) for one row block of the UNSCALED kernel (single lengthscale), float64: a_rows [r, d + 2], b_ [d + 2, m].
    ``row0``: index of the first row when the block sits on the diagonal of a square matrix (exact zeros of sq_dist there)."""
    res = a_rows @ b_
    if row0 is not None:
        idx = torch.arange(a_rows.shape[0], device=res.device)
        res[idx, idx + row0] = 0.0
    res.clamp_min_(0.0)
    if kind == "rbf":
        K = torch.exp(-0.5 * res)
        dl = K * res / lengthscale                       # d/dl exp(-d^2 / (2 l^2)) = K d^2 / l^3 = K S / l
        return K, dl
    nu = {"matern12": 0.5, "matern32": 1.5, "matern52": 2.5}[kind]
    r = res.clamp_min_(1e-30).sqrt_().mul_(math.sqrt(2 * nu))
    ex = torch.exp(-r)
    if nu == 0.5:
        return ex, r * ex / lengthscale
    if nu == 1.5:
        return (1.0 + r) * ex, r * r * ex / lengthscale
    return (1.0 + r + r * r / 3.0) * ex, (r * r / 3.0) * (1.0 + r) * ex / lengthscale


def bilinear_forms(kind, X, lengthscale: float, outputscale: float, Lv: torch.Tensor, Rv: torch.Tensor, device, block: int = 4096):
    """Per column c: L_c^T (d K_hat / d theta) R_c for theta in (lengthscale, outputscale, noise), K_hat = outputscale K + noise I, in
    float64 by dense row blocks.  Lv, Rv: [n, c] float64 on the device.  Returns three [c] tensors."""
    n = X.shape[0]
    a_, b_ = _scaled_operands(kind, X, lengthscale, device)
    c = Lv.shape[1]
    g_l = torch.zeros(c, dtype=torch.float64, device=device)
    g_o = torch.zeros(c, dtype=torch.float64, device=device)
    for a in range(0, n, block):
        e = min(n, a + block)
        K, dl = kernel_rows_and_dl(kind, a_[a:e], b_, lengthscale, row0=a)
        g_o += (Lv[a:e] * (K @ Rv)).sum(0)
        g_l += (Lv[a:e] * (dl @ Rv)).sum(0)
        del K, dl
    return outputscale * g_l, g_o, (Lv * Rv).sum(0)


def cross_rows(kind, X, Xs, lengthscale: float, outputscale: float, device):
    """outputscale * K(Xs, X) in float64 ([ns, n]); both clouds get the TRAINING cloud's centring, as the reference's kernel call on the
    concatenated inputs does up to a common shift (stationary kernels: any common shift is exact)."""
    x = X.to(device=device, dtype=torch.float64)
    xs = Xs.to(device=device, dtype=torch.float64)
    mu = x.mean(-2, keepdim=True)
    x, xs = (x - mu) / lengthscale, (xs - mu) / lengthscale
    d2 = (xs.pow(2).sum(-1, keepdim=True) + x.pow(2).sum(-1).unsqueeze(0) - 2.0 * xs @ x.t()).clamp_min_(0.0)
    if kind == "rbf":
        return outputscale * torch.exp(-0.5 * d2)
    nu = {"matern12": 0.5, "matern32": 1.5, "matern52": 2.5}[kind]
    r = d2.clamp_min_(1e-30).sqrt_().mul_(math.sqrt(2 * nu))
    ex = torch.exp(-r)
    k = ex if nu == 0.5 else ((1.0 + r) * ex if nu == 1.5 else (1.0 + r + r * r / 3.0) * ex)
    return outputscale * k


class DenseGP:
    """K_hat = outputscale K + noise I factorised once; log det, solves, the exact posterior."""

    def __init__(self, kind, X, y, lengthscale, outputscale, noise, device, block=4096):
        self.kind, self.X, self.ls, self.os, self.s2, self.dev, self.block = kind, X, float(lengthscale), float(outputscale), float(noise), device, block
        self.L = cholesky_inplace_(dense_khat(kind, X, lengthscale, outputscale, noise, device, block), block)
        self.logdet = logdet_from_factor(self.L)
        self.y = y.to(device=device, dtype=torch.float64).reshape(-1, 1)
        self.alpha = self.solve(self.y)
        self.inv_quad = float((self.alpha * self.y).sum())

    def solve(self, b):
        return solve_with_factor(self.L, b.to(device=self.dev, dtype=torch.float64), self.block)

    def posterior(self, Xs):
        """(mean, variance of f, K_*X) at the test points, float64 (zero prior mean)."""
        ks = cross_rows(self.kind, self.X, Xs, self.ls, self.os, self.dev)          # [ns, n]
        mean = (ks @ self.alpha).squeeze(-1)
        v = self.solve(ks.t().contiguous())                                           # [n, ns]
        var = self.os - (ks * v.t()).sum(-1)
        return mean, var

    def free(self):
        self.L = None
        torch.cuda.empty_cache()
