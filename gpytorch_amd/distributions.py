"""``MultivariateNormal`` over a LinearOperator covariance -- the call site of ``inv_quad_logdet``
(``gpytorch/distributions/multivariate_normal.py:221-252``) and of ``diagonal`` for variances
(``:361-382``)."""
from __future__ import annotations

import math
import warnings

import torch

from . import settings
from .linear_cg import NumericalWarning
from .operators import LinearOperator, psd_safe_cholesky, to_dense, to_linear_operator


class MultivariateNormal:
    def __init__(self, mean: torch.Tensor, covariance_matrix, validate_args=False):
        self.loc = mean
        self._covar = covariance_matrix
        self.islazy = isinstance(covariance_matrix, LinearOperator)
        if covariance_matrix.shape[-1] != mean.shape[-1] or covariance_matrix.shape[-2] != mean.shape[-1]:
            raise RuntimeError(f"mean {tuple(mean.shape)} and covariance {tuple(covariance_matrix.shape)} sizes do not match")

    @property
    def mean(self):
        return self.loc

    @property
    def event_shape(self):
        return self.loc.shape[-1:]

    @property
    def batch_shape(self):
        return self.loc.shape[:-1]

    @property
    def lazy_covariance_matrix(self):
        return to_linear_operator(self._covar)

    @property
    def covariance_matrix(self):
        return to_dense(self._covar)

    def log_prob(self, value: torch.Tensor) -> torch.Tensor:
        """multivariate_normal.py:221-252."""
        mean, covar = self.loc, self.lazy_covariance_matrix
        diff = value - mean
        if settings.fast_computations.log_prob.off() and not hasattr(covar, "kernel_op") and not hasattr(covar, "ops"):
            Lc = psd_safe_cholesky(covar.to_dense())
            sol = torch.cholesky_solve(diff.unsqueeze(-1), Lc).squeeze(-1)
            return -0.5 * ((diff * sol).sum(-1) + 2 * Lc.diagonal(dim1=-2, dim2=-1).log().sum(-1) + diff.size(-1) * math.log(2 * math.pi))
        covar = covar.evaluate_kernel()
        inv_quad, logdet = covar.inv_quad_logdet(inv_quad_rhs=diff.unsqueeze(-1), logdet=True)
        return -0.5 * sum([inv_quad, logdet, diff.size(-1) * math.log(2 * math.pi)])

    @property
    def variance(self):
        """multivariate_normal.py:361-382 (clamped at settings.min_variance)."""
        variance = self.lazy_covariance_matrix.diagonal(dim1=-1, dim2=-2)
        min_variance = settings.min_variance.value(variance.dtype)
        if variance.lt(min_variance).any():
            warnings.warn(
                f"Negative variance values detected. This is likely due to numerical instabilities. "
                f"Rounding negative variances up to {min_variance}.",
                NumericalWarning,
            )
            variance = variance.clamp_min(min_variance)
        return variance

    @property
    def stddev(self):
        return self.variance.sqrt()

    def confidence_region(self):
        std2 = self.stddev.mul(2)
        return self.mean.sub(std2), self.mean.add(std2)

    def __add__(self, other):
        if isinstance(other, MultivariateNormal):
            return MultivariateNormal(self.mean + other.mean, self.lazy_covariance_matrix + other.lazy_covariance_matrix)
        return MultivariateNormal(self.mean + other, self._covar)

    def rsample(self, sample_shape=torch.Size(), base_samples=None):
        """mu + R eps with R a root of the covariance (``multivariate_normal.py:254-320``): without ``base_samples`` through
        ``covar.zero_mean_mvn_samples`` (Cholesky root for small operators, the matrix-free Lanczos root above
        ``max_cholesky_size`` or under ``settings.fast_pred_samples``); with ``base_samples`` ([*sample_shape, n] or
        [*sample_shape, rank]) they are pushed through the root, truncated to its rank if it is low-rank."""
        covar = self.lazy_covariance_matrix
        sample_shape = torch.Size(sample_shape)
        if base_samples is None:
            num = sample_shape.numel() or 1
            res = covar.zero_mean_mvn_samples(num) + self.loc.unsqueeze(0)
            return res.view(sample_shape + self.loc.shape)
        root = covar.root_decomposition().root
        if self.loc.shape != base_samples.shape[-self.loc.dim():] and root.shape[-1] < base_samples.shape[-1]:
            raise RuntimeError("The size of base_samples (minus sample shape dimensions) should agree with the size of self.loc. "
                               f"Expected ...{self.loc.shape} but got {base_samples.shape}")
        sample_shape = base_samples.shape[: base_samples.dim() - self.loc.dim()]
        eps = base_samples.reshape(-1, base_samples.shape[-1])[:, : root.shape[-1]]      # low-rank root: the first `rank` draws
        res = eps @ root.mT + self.loc
        return res.view(sample_shape + self.loc.shape)

    sample = rsample
