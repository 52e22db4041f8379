"""Oracle: multitask exact GP (Kronecker K_XX (x) K_TT + I (x) D), dense float64.  Test infrastructure only.

Restates ``gpytorch/kernels/multitask_kernel.py:46-54`` (K = K_XX (x) K_TT), ``kernels/index_kernel.py:91-99``
(K_TT = B B^T + diag(v)), ``likelihoods/multitask_gaussian_likelihood.py:118-154`` (noise I_n (x) (D_T + s2 I_T))
and the interleaved layout of ``distributions/multitask_multivariate_normal.py:66-70`` (row = i*T + tau, which is
exactly ``torch.kron(K_XX, K_TT)``).  Ground truth by Cholesky, as in
``test/examples/test_kronecker_multitask_gp_regression.py:55-90``.
"""
from __future__ import annotations

import torch

from . import kernels as K_
from .exact_gp import dense_log_prob


def task_covar(Bf, v):
    return Bf @ Bf.t() + torch.diag(v)


def khat(kind, X, ls, os_, Bf, v, task_noise):
    n = X.shape[-2]
    Kxx = K_.kernel_matrix(kind, X, X, ls, os_, x1_eq_x2=True)
    return torch.kron(Kxx, task_covar(Bf, v)) + torch.diag(task_noise.repeat(n))


def dense_mll(kind, X, Y, ls, os_, Bf, v, task_noise, mean=0.0):
    return dense_log_prob(khat(kind, X, ls, os_, Bf, v, task_noise), (Y - mean).reshape(-1)) / Y.numel()


def dense_mll_and_grads(kind, X, Y, ls, os_, Bf, v, task_noise, mean=0.0):
    p = [torch.as_tensor(a, dtype=X.dtype).clone().requires_grad_(True) for a in (ls, Bf, v, task_noise)]
    val = dense_mll(kind, X, Y, p[0], os_, p[1], p[2], p[3], mean)
    g = torch.autograd.grad(val, p)
    return val.detach(), [a.detach() for a in g]


def dense_posterior(kind, X, Y, Xs, ls, os_, Bf, v, task_noise, mean=0.0, noise=True):
    n, T = Y.shape
    Kh = khat(kind, X, ls, os_, Bf, v, task_noise)
    Ktt = task_covar(Bf, v)
    Ksx = torch.kron(K_.kernel_matrix(kind, Xs, X, ls, os_, x1_eq_x2=False), Ktt)
    Lc = torch.linalg.cholesky(Kh)
    alpha = torch.cholesky_solve((Y - mean).reshape(-1, 1), Lc)
    mu = (Ksx @ alpha).reshape(-1, T) + mean
    prior = (os_ * torch.ones(Xs.shape[-2], dtype=X.dtype)).unsqueeze(-1) * Ktt.diagonal().unsqueeze(0)
    w = torch.linalg.solve_triangular(Lc, Ksx.t(), upper=False)
    var = prior - w.pow(2).sum(0).reshape(-1, T)
    if noise:
        var = var + task_noise.unsqueeze(0)
    return mu, var
