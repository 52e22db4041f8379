"""``ExactMarginalLogLikelihood``: caller glue around the hot path, kept API-compatible with
``gpytorch/mlls/exact_marginal_log_likelihood.py:36-89`` (north_star: "keeping the ... ExactMarginalLogLikelihood API
surface").  When the real ``gpytorch`` is importable the dual-mode module (:mod:`gpytorch_amd.dropin`) uses ITS class
unchanged; this stand-in only assembles

    ( log N(y | mu, K_hat)  +  added-loss terms  +  sum of prior log-densities ) / n

where the first term is :meth:`MultivariateNormal.log_prob` -> ``inv_quad_logdet`` on the fused operator.
"""
from __future__ import annotations

import torch

from . import settings
from .distributions import MultivariateNormal
from .likelihoods import _GaussianLikelihoodBase
from .module import Module


class MarginalLogLikelihood(Module):
    def __init__(self, likelihood, model):
        super().__init__()
        self.likelihood = likelihood
        self.model = model


def _observed_only(marginal: MultivariateNormal, target: torch.Tensor):
    """observation_nan_policy "mask" (reference :68-77): condition on the observed entries only.  The fused operators
    restrict themselves to a subset of the points (no dense masking operator is needed)."""
    keep = ~torch.isnan(target.reshape(-1, *marginal.event_shape)).any(dim=0)
    idx = keep.reshape(-1).nonzero().squeeze(-1)
    covar = marginal.lazy_covariance_matrix
    sub = covar.restrict(idx) if hasattr(covar, "restrict") else covar[idx][:, idx]
    return MultivariateNormal(marginal.mean[..., idx], sub), target[..., idx]


class ExactMarginalLogLikelihood(MarginalLogLikelihood):
    def __init__(self, likelihood, model):
        if not isinstance(likelihood, _GaussianLikelihoodBase):
            raise RuntimeError("Likelihood must be Gaussian for exact inference")
        super().__init__(likelihood, model)

    def _other_terms(self, ndim: int, params):
        """Added-loss terms and prior log-densities, each reduced to the batch shape (``ndim`` leading dims) of the MLL."""
        extra = [term.loss(*params) for term in self.model.added_loss_terms()]
        for _, module, prior, closure, _ in self.model.named_priors():
            lp = prior.log_prob(closure(module))
            extra.append(lp.reshape(*lp.shape[:ndim], -1).sum(dim=-1))
        return extra

    def forward(self, function_dist, target, *params, **kwargs):
        if not isinstance(function_dist, MultivariateNormal):
            raise RuntimeError("ExactMarginalLogLikelihood can only operate on Gaussian random variables")
        marginal = self.likelihood(function_dist, *params, **kwargs)
        policy = settings.observation_nan_policy.value()
        if policy == "fill":
            raise ValueError("NaN observation policy 'fill' is not supported by ExactMarginalLogLikelihood!")
        if policy == "mask":
            marginal, target = _observed_only(marginal, target)
        total = marginal.log_prob(target)
        for term in self._other_terms(total.dim(), params):
            total = total + term
        return total / function_dist.event_shape.numel()
