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


MASK_DENSE_LIMIT = 8192    # structured operators without a point-restricted form are masked densely up to this many rows


def _observed_only(marginal: MultivariateNormal, target: torch.Tensor):
    """observation_nan_policy "mask" (reference :68-77): condition on the observed entries only -- an entry of the EVENT (a point, or a (point, task)
    pair of a multitask distribution, flattened in the interleaved order of its covariance) counts as observed if every batch member observes it.
    The fused operators restrict themselves to a subset of the points (no masking operator is needed); an operator whose structure an arbitrary subset
    breaks (Kronecker multitask with some tasks of a point missing) is masked on its dense form, as the reference's ``MaskedLinearOperator``
    ends up doing inside a Cholesky-sized solve."""
    event = marginal.event_shape
    keep = ~torch.isnan(target.reshape(-1, *event)).any(dim=0)
    idx = keep.reshape(-1).nonzero().squeeze(-1)
    batch = target.shape[: target.dim() - len(event)]
    mean = marginal.mean.reshape(*marginal.mean.shape[: marginal.mean.dim() - len(event)], -1)[..., idx]
    target = target.reshape(*batch, -1)[..., idx]
    covar = marginal.lazy_covariance_matrix
    if hasattr(covar, "restrict") and len(event) == 1:
        sub = covar.restrict(idx)
    else:
        if covar.shape[-1] > MASK_DENSE_LIMIT:
            raise NotImplementedError(f"observation_nan_policy('mask') on a {type(covar).__name__} of {covar.shape[-1]} rows: no point-restricted form")
        from .operators import DenseLinearOperator, to_dense

        dense = to_dense(covar)
        sub = DenseLinearOperator(dense[..., idx, :][..., :, idx])
    return MultivariateNormal(mean, sub), target


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
