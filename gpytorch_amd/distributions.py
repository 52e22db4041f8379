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
        if mean.dim() > 0 and (covariance_matrix.shape[-1] != mean.shape[-1] or covariance_matrix.shape[-2] != mean.shape[-1]):
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
            return self.__class__(self.mean + other.mean, self.lazy_covariance_matrix + other.lazy_covariance_matrix)
        return self.__class__(self.mean + other, self._covar)

    def __radd__(self, other):
        return self if (not isinstance(other, MultivariateNormal) and other == 0) else self.__add__(other)

    def __mul__(self, other):
        """multivariate_normal.py:436-441: a scalar c scales the mean by c and the covariance by c^2."""
        if not isinstance(other, (int, float)):
            raise RuntimeError("Can only multiply by scalars")
        if other == 1:
            return self
        covar = self._covar.mul(other ** 2) if self.islazy else self._covar * (other ** 2)
        return self.__class__(self.mean * other, covar)

    def __truediv__(self, other):
        return self.__mul__(1.0 / other)

    # ---- the torch.distributions surface the reference inherits (multivariate_normal.py:30-120, 322-393) ----------------------------------
    @property
    def _unbroadcasted_scale_tril(self):
        if getattr(self, "_scale_tril", None) is None:
            self._scale_tril = psd_safe_cholesky(self.covariance_matrix)
        return self._scale_tril

    @property
    def scale_tril(self):
        return self._unbroadcasted_scale_tril

    def entropy(self):
        """H = n / 2 (1 + log 2 pi) + 1/2 log|Sigma|  (log|Sigma| from the Cholesky factor, as torch's MultivariateNormal computes it)."""
        n = self.loc.shape[-1]
        return 0.5 * n * (1.0 + math.log(2 * math.pi)) + self._unbroadcasted_scale_tril.diagonal(dim1=-2, dim2=-1).log().sum(-1)

    def _rewrap(self, dense):
        """A dense covariance in the form this distribution was built with (``islazy`` is kept across expand / unsqueeze / indexing)."""
        from .operators import DenseLinearOperator

        return DenseLinearOperator(dense) if self.islazy else dense

    def expand(self, batch_size):
        """multivariate_normal.py:172-186: both parameters expanded to a larger batch shape (the Cholesky factor, if it was built, with them)."""
        batch = torch.Size(batch_size)
        n = self.loc.shape[-1]
        new = self.__class__(self.loc.expand(*batch, n), self._rewrap(self.covariance_matrix.expand(*batch, n, n)))
        if getattr(self, "_scale_tril", None) is not None:
            new._scale_tril = self._scale_tril.expand(*batch, n, n)
        return new

    def unsqueeze(self, dim: int):
        """multivariate_normal.py:188-214: a new batch dimension at ``dim`` (counted among the batch dimensions; negative: from their end)."""
        nb = len(self.batch_shape)
        if dim > nb or dim < -nb - 1:
            raise IndexError(f"Dimension out of range (expected to be in range of [{-nb - 1}, {nb}], but got {dim}).")
        if dim < 0:
            dim = nb + dim + 1
        new = self.__class__(self.loc.unsqueeze(dim), self._rewrap(self.covariance_matrix.unsqueeze(dim)))
        if getattr(self, "_scale_tril", None) is not None:
            new._scale_tril = self._scale_tril.unsqueeze(dim)
        return new

    def __getitem__(self, idx):
        """multivariate_normal.py:395-434: the distribution of the indexed random variable.  Indices that touch only batch dimensions index the
        operator; an index on the event dimension selects rows AND columns of the covariance (an integer there leaves the variances of the
        selected entry over what was the last batch dimension: a diagonal covariance)."""
        from .operators import DiagLinearOperator

        idx = idx if isinstance(idx, tuple) else (idx,)
        nd = self.loc.dim()
        has_ellipsis = any(i is Ellipsis for i in idx)
        if len(idx) > nd and has_ellipsis:
            idx = tuple(i for i in idx if i is not Ellipsis)
            if len(idx) < nd:
                raise IndexError("Multiple ambiguous ellipsis in index!")
        rest, last = idx[:-1], idx[-1]
        new_mean = self.loc[idx]
        dense = self.covariance_matrix
        if len(idx) <= nd - 1 and not any(i is Ellipsis for i in rest) and last is not Ellipsis:
            new_cov = self._rewrap(dense[idx])                                      # batch dimensions only
        elif len(idx) > nd:
            raise IndexError(f"Index {idx} has too many dimensions")
        elif isinstance(last, int):
            var = dense.diagonal(dim1=-1, dim2=-2)[(*rest, last)]
            new_cov = DiagLinearOperator(var) if var.dim() > 0 else var             # (every dimension indexed away: a scalar variance)
        elif isinstance(last, slice):
            new_cov = self._rewrap(dense[(*rest, last, last)])
        elif last is Ellipsis:
            new_cov = self._rewrap(dense[rest])
        else:
            new_cov = self._rewrap(dense[(*rest, last, slice(None, None, None))][..., last])
        return self.__class__(new_mean, new_cov)

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
