"""Oracle: covariance functions (test infrastructure; never imported by the product).

Restates, in plain CPU PyTorch, the arithmetic of
  * ``gpytorch/kernels/kernel.py:26-49``   (sq_dist: mean-centred Gram trick, clamp>=0, zero diag)
  * ``gpytorch/kernels/kernel.py:52-60``   (dist: cdist clamp 1e-15 / sqrt(clamp(sq_dist,1e-30)))
  * ``gpytorch/functions/rbf_covariance.py:14-23``   (RBF value and dK/dl)
  * ``gpytorch/functions/matern_covariance.py:18-50`` (Matern nu=1/2,3/2,5/2 value and dK/dl)
  * ``gpytorch/kernels/scale_kernel.py:108-118``      (outputscale * K)
  * ``gpytorch/kernels/keops/rbf_kernel.py:12-15`` and ``keops/matern_kernel.py:13-30``
    (the direct pairwise-difference forms used by the matrix-free reference path)
"""
from __future__ import annotations

import math

import torch


def sq_dist(x1: torch.Tensor, x2: torch.Tensor, x1_eq_x2: bool = False) -> torch.Tensor:
    """kernels/kernel.py:26-49."""
    adjustment = x1.mean(-2, keepdim=True)
    x1 = x1 - adjustment
    x1_norm = x1.pow(2).sum(dim=-1, keepdim=True)
    x1_pad = torch.ones_like(x1_norm)
    if x1_eq_x2:
        x2, x2_norm, x2_pad = x1, x1_norm, x1_pad
    else:
        x2 = x2 - adjustment
        x2_norm = x2.pow(2).sum(dim=-1, keepdim=True)
        x2_pad = torch.ones_like(x2_norm)
    x1_ = torch.cat([-2.0 * x1, x1_norm, x1_pad], dim=-1)
    x2_ = torch.cat([x2, x2_pad, x2_norm], dim=-1)
    res = x1_.matmul(x2_.transpose(-2, -1))
    if x1_eq_x2:
        res.diagonal(dim1=-2, dim2=-1).fill_(0)
    return res.clamp_min_(0)


def dist(x1: torch.Tensor, x2: torch.Tensor, x1_eq_x2: bool = False) -> torch.Tensor:
    """kernels/kernel.py:52-60."""
    if not x1_eq_x2:
        return torch.cdist(x1, x2).clamp_min(1e-15)
    return sq_dist(x1, x2, x1_eq_x2=True).clamp_min_(1e-30).sqrt_()


def sq_dist_direct(x1: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
    """Pairwise-difference squared distance (keops/rbf_kernel.py:12-15 form)."""
    return (x1.unsqueeze(-2) - x2.unsqueeze(-3)).pow(2).sum(-1)


def rbf(x1, x2, lengthscale, x1_eq_x2=None, direct=False):
    """functions/rbf_covariance.py:14-19: exp(-0.5 * sq_dist(x1/l, x2/l))."""
    if x1_eq_x2 is None:
        x1_eq_x2 = x1.shape == x2.shape and torch.equal(x1, x2)
    x1_ = x1.div(lengthscale)
    x2_ = x2.div(lengthscale)
    d2 = sq_dist_direct(x1_, x2_) if direct else sq_dist(x1_, x2_, x1_eq_x2)
    return d2.div(-2.0).exp()


def rbf_dl(x1, x2, lengthscale, x1_eq_x2=None):
    """functions/rbf_covariance.py:21: dK/dl = sq_dist * K / l (single lengthscale)."""
    if x1_eq_x2 is None:
        x1_eq_x2 = x1.shape == x2.shape and torch.equal(x1, x2)
    d2 = sq_dist(x1.div(lengthscale), x2.div(lengthscale), x1_eq_x2)
    return d2 * d2.div(-2.0).exp() / lengthscale


def matern(x1, x2, lengthscale, nu, x1_eq_x2=None, direct=False):
    """functions/matern_covariance.py:18-50 (dense form) or keops/matern_kernel.py:13-30 (direct)."""
    if x1_eq_x2 is None:
        x1_eq_x2 = x1.shape == x2.shape and torch.equal(x1, x2)
    mean = x1.mean(dim=-2, keepdim=True)
    x1_ = (x1 - mean).div(lengthscale)
    x2_ = (x2 - mean).div(lengthscale)
    if direct:
        r = (sq_dist_direct(x1_, x2_) + 1e-20).sqrt()
    else:
        r = dist(x1_, x2_, x1_eq_x2)
    s = r * math.sqrt(2 * nu)
    e = torch.exp(-s)
    if nu == 0.5:
        return e
    if nu == 1.5:
        return (s + 1) * e
    if nu == 2.5:
        return (1 + s + s.pow(2) / 3) * e
    raise RuntimeError("nu expected to be 0.5, 1.5, or 2.5")


def matern_dl(x1, x2, lengthscale, nu, x1_eq_x2=None):
    """functions/matern_covariance.py:30,39,47: dK/dl for a single lengthscale."""
    if x1_eq_x2 is None:
        x1_eq_x2 = x1.shape == x2.shape and torch.equal(x1, x2)
    mean = x1.mean(dim=-2, keepdim=True)
    s = dist((x1 - mean).div(lengthscale), (x2 - mean).div(lengthscale), x1_eq_x2) * math.sqrt(2 * nu)
    e = torch.exp(-s)
    if nu == 0.5:
        return s / lengthscale * e
    if nu == 1.5:
        return s.pow(2) / lengthscale * e
    if nu == 2.5:
        return (s + 1) * (s.pow(2) / 3) * e / lengthscale
    raise RuntimeError("nu expected to be 0.5, 1.5, or 2.5")


def rq(x1, x2, lengthscale, alpha, x1_eq_x2=None, direct=False):
    """kernels/rq_kernel.py:60-74: (1 + sq_dist(x1 / l, x2 / l) / (2 alpha))^-alpha."""
    if x1_eq_x2 is None:
        x1_eq_x2 = x1.shape == x2.shape and torch.equal(x1, x2)
    x1_, x2_ = x1.div(lengthscale), x2.div(lengthscale)
    d2 = sq_dist_direct(x1_, x2_) if direct else sq_dist(x1_, x2_, x1_eq_x2)
    return (1 + d2.div(2 * alpha)).pow(-alpha)


def periodic(x1, x2, lengthscale, period_length):
    """kernels/periodic_kernel.py:125-142: exp(-2 sum_q sin^2(pi (x1_q - x2_q) / p_q) / l_q); lengthscale / period_length [1, d or 1].
    (The reference takes the per-dimension |x1_q - x2_q| through covar_dist(last_dim_is_batch=True); sin^2 is even, so the signed
    difference gives the same value.)"""
    diff = math.pi * (x1.unsqueeze(-2) - x2.unsqueeze(-3)) / period_length
    return torch.exp(-2.0 * (diff.sin().pow(2) / lengthscale).sum(-1))


KINDS = {"rbf": None, "matern12": 0.5, "matern32": 1.5, "matern52": 2.5}


def kernel_matrix(kind, x1, x2, lengthscale, outputscale=1.0, x1_eq_x2=None, direct=False):
    """ScaleKernel(kind)(x1, x2).to_dense()  (kernels/scale_kernel.py:108-118)."""
    if kind == "rbf":
        k = rbf(x1, x2, lengthscale, x1_eq_x2, direct)
    else:
        k = matern(x1, x2, lengthscale, KINDS[kind], x1_eq_x2, direct)
    return k * outputscale


def kernel_matmul_chunked(kind, x1, x2, lengthscale, outputscale, rhs, chunk=4096):
    """Matrix-free K @ rhs: the reference's own chunked path
    (lazy/lazy_evaluated_kernel_tensor.py:245-275: split x1 rows, build K_chunk, matmul, cat).
    Matern centres with the mean of the FULL x1 (as the un-chunked kernel call would)."""
    outs = []
    if kind != "rbf":
        mean = x1.mean(dim=-2, keepdim=True)
        x1, x2 = x1 - mean, x2 - mean
    for s in range(0, x1.shape[-2], chunk):
        xc = x1[s : s + chunk]
        if kind == "rbf":
            kc = sq_dist_direct(xc / lengthscale, x2 / lengthscale).div(-2.0).exp()
        else:
            nu = KINDS[kind]
            r = (sq_dist_direct(xc / lengthscale, x2 / lengthscale) + 1e-20).sqrt() * math.sqrt(2 * nu)
            e = torch.exp(-r)
            kc = e if nu == 0.5 else ((r + 1) * e if nu == 1.5 else (1 + r + r.pow(2) / 3) * e)
        outs.append((kc * outputscale) @ rhs)
    return torch.cat(outs, dim=-2)


def kernel_matmul_rows(kind, x, rows, lengthscale, outputscale, rhs, chunk=128):
    """Rows ``rows`` of ``K(x, x) @ rhs`` through the reference's DEFAULT dense formulas, one row chunk at a time:
    RBF via the mean-centred Gram-trick ``sq_dist`` (kernels/kernel.py:26-49, functions/rbf_covariance.py:14-19),
    Matern via ``dist`` = ``torch.cdist`` for x1 != x2 (kernels/kernel.py:52-60, functions/matern_covariance.py:18-50).
    In float64 the Gram-trick cancellation is ~1e-15, so this is the exact value to the tolerance of any float32 path,
    while the temporaries stay at chunk x n (the pairwise-difference form needs chunk x n x d)."""
    outs = []
    xr = x[rows]
    if kind != "rbf":
        mean = x.mean(dim=-2, keepdim=True)  # matern_kernel.py:94: centred by the mean of the full x1
        xr, x = xr - mean, x - mean
    for s in range(0, xr.shape[-2], chunk):
        xc = xr[s : s + chunk]
        if kind == "rbf":
            kc = sq_dist(xc / lengthscale, x / lengthscale, False).div_(-2.0).exp_()
        else:
            nu = KINDS[kind]
            r = torch.cdist(xc / lengthscale, x / lengthscale).clamp_min_(1e-15).mul_(math.sqrt(2 * nu))
            e = torch.exp(-r)
            kc = e if nu == 0.5 else (e.mul_(r + 1) if nu == 1.5 else e.mul_(1 + r + r.pow(2) / 3))
        outs.append((kc @ rhs) * outputscale)
    return torch.cat(outs, dim=-2)
