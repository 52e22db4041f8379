"""Oracle: the two example-notebook trainings whose PRINTED OUTPUTS the reference ships (``tests/golden/make_notebook_golden.py``), restated on
this package's dense float64 formulas.  Test infrastructure only.

The printed numbers were produced by the complete reference stack -- ``gpytorch`` on the real ``linear_operator`` -- so a trajectory of this
module that lands on them pins, against the reference ITSELF rather than against an independent dense evaluation:

* the kernel formulas of ``oracle/kernels.py`` inside an MLL (``kernels/rbf_kernel.py``, ``kernels/scale_kernel.py:108-118``);
* the MLL assembly of ``oracle/exact_gp.py::dense_log_prob`` (``distributions/multivariate_normal.py:221-252``,
  ``mlls/exact_marginal_log_likelihood.py:83-89``: divide by the number of data) AND ITS GRADIENT with respect to every hyper-parameter: 50-100 Adam
  steps amplify any gradient error into the printed third decimal;
* the raw-parameter transforms (``constraints/constraints.py``: ``Positive`` = softplus, ``GreaterThan(1e-4)`` = softplus + 1e-4 for the noises,
  ``likelihoods/noise_models.py:40-50``);
* the Hadamard multitask covariance (``kernels/index_kernel.py:91-112``: (B B^T + diag(v))[i, i'] times the data kernel), the per-task noise of
  ``likelihoods/hadamard_gaussian_likelihood.py:88-111`` and the fixed + learned noise of ``likelihoods/gaussian_likelihood.py:283-296``;
* and, through ``bbmm_loss``, the VALUE that this package's mBCG + stochastic-Lanczos-quadrature restatement (``linear_cg.py``, ``slq.py``)
  returns for the same matrices: with the n unit vectors as probes the quadrature is the exact trace, so the BBMM route must print the same
  number.  (Iteration-level parity of mBCG stays unpinned: the reference publishes no iterates.)
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import kernels as K_
from .exact_gp import dense_log_prob
from .slq import inv_quad_logdet

NOISE_FLOOR = 1e-4     # GreaterThan(1e-4): likelihoods/noise_models.py:44


def hadamard_parameters(covar_factor, raw_var, per_task_noise: bool, dtype=torch.float64):
    """Initial raw parameters of the notebook's model: everything at raw 0 (softplus -> 0.693) except the IndexKernel draws."""
    p = {
        "raw_noise": torch.zeros(2 if per_task_noise else 1), "constant": torch.zeros(()), "raw_lengthscale": torch.zeros(1, 1),
        "covar_factor": torch.as_tensor(covar_factor).clone(), "raw_var": torch.as_tensor(raw_var).clone(),
    }
    return {k: v.to(dtype).requires_grad_(True) for k, v in p.items()}


def hadamard_khat_and_residual(p, x, i, y):
    """K_hat = RBF(x / l) o (B B^T + diag(softplus(raw_var)))[i, i'] + noise (shared, or looked up at the task index), residual y - constant."""
    n = x.shape[0]
    kx = K_.rbf(x.unsqueeze(-1), x.unsqueeze(-1), F.softplus(p["raw_lengthscale"]), x1_eq_x2=True)
    ktt = p["covar_factor"] @ p["covar_factor"].t() + torch.diag(F.softplus(p["raw_var"]))
    noise = F.softplus(p["raw_noise"]) + NOISE_FLOOR
    diag = noise[i] if noise.numel() == 2 else noise.expand(n)
    return kx * ktt[i][:, i] + torch.diag(diag), y - p["constant"]


def hadamard_loss(p, x, i, y):
    khat, r = hadamard_khat_and_residual(p, x, i, y)
    return -dense_log_prob(khat, r) / y.shape[0]


def classification_parameters(members: int, dtype=torch.float64):
    p = {"raw_noise": torch.zeros(members, 1), "constant": torch.zeros(members), "raw_lengthscale": torch.zeros(members, 1, 1), "raw_outputscale": torch.zeros(members)}
    return {k: v.to(dtype).requires_grad_(True) for k, v in p.items()}


def classification_khat_and_residual(p, X, targets, fixed_noise):
    """One K_hat per member (class): outputscale_c RBF(X / l_c) + diag(fixed_noise_c) + learned noise_c."""
    ks = []
    for c in range(targets.shape[0]):
        kc = K_.kernel_matrix("rbf", X, X, F.softplus(p["raw_lengthscale"][c]), F.softplus(p["raw_outputscale"][c]), x1_eq_x2=True)
        ks.append(kc + torch.diag(fixed_noise[c] + F.softplus(p["raw_noise"][c]) + NOISE_FLOOR))
    return torch.stack(ks), targets - p["constant"].unsqueeze(-1)


def classification_loss(p, X, targets, fixed_noise):
    """Minus the SUM over members of the per-datum MLL (the notebook's ``-mll(output, targets).sum()``)."""
    khat, r = classification_khat_and_residual(p, X, targets, fixed_noise)
    return -(dense_log_prob(khat, r) / targets.shape[-1]).sum()


def bbmm_loss(khat: torch.Tensor, r: torch.Tensor) -> torch.Tensor:
    """The same loss through this package's BBMM restatement: mBCG on [e_1 .. e_n | r] run to the end, log-det from the n tridiagonals
    (exact trace: the probes are the unit vectors), y^T K^-1 y from the last column.  Batched K_hat: summed over members."""
    if khat.dim() == 3:
        return sum(bbmm_loss(k, rr) for k, rr in zip(khat, r))
    n = khat.shape[-1]
    iq, ld = inv_quad_logdet(lambda v: khat @ v, n, r.unsqueeze(-1), torch.eye(n, dtype=khat.dtype), tolerance=1e-10, max_iter=4 * n, max_tridiag_iter=n)
    return 0.5 * (iq.sum() + ld + n * math.log(2 * math.pi)) / n


def adam_trajectory(loss_fn, params: dict, steps: int, lr: float = 0.1, snapshot_at=()):
    """``steps`` iterations of the notebooks' loop (zero_grad, loss, backward, step).  Returns the loss BEFORE each step (what the notebooks print)
    and detached copies of the parameters as they stood when the iterations in ``snapshot_at`` (1-based) evaluated their loss."""
    opt = torch.optim.Adam(list(params.values()), lr=lr)
    losses, snaps = [], {}
    for it in range(1, steps + 1):
        opt.zero_grad()
        loss = loss_fn(params)
        loss.backward()
        losses.append(float(loss.detach()))
        if it in snapshot_at:
            snaps[it] = {k: v.detach().clone() for k, v in params.items()}
        opt.step()
    return losses, snaps
