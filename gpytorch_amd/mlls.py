"""``ExactMarginalLogLikelihood`` -- same assembly as ``gpytorch/mlls/exact_marginal_log_likelihood.py:54-89``:
likelihood(f) -> log_prob(y) -> + added-loss / prior terms -> / n."""
from __future__ import annotations

from .distributions import MultivariateNormal
from .likelihoods import _GaussianLikelihoodBase
from .module import Module


class MarginalLogLikelihood(Module):
    def __init__(self, likelihood, model):
        super().__init__()
        self.likelihood = likelihood
        self.model = model


class ExactMarginalLogLikelihood(MarginalLogLikelihood):
    def __init__(self, likelihood, model):
        if not isinstance(likelihood, _GaussianLikelihoodBase):
            raise RuntimeError("Likelihood must be Gaussian for exact inference")
        super().__init__(likelihood, model)

    def _add_other_terms(self, res, params):
        for added_loss_term in self.model.added_loss_terms():
            res = res.add(added_loss_term.loss(*params))
        res_ndim = res.ndim
        for name, module, prior, closure, _ in self.model.named_priors():
            prior_term = prior.log_prob(closure(module))
            res = res + prior_term.view(*prior_term.shape[:res_ndim], -1).sum(dim=-1)
        return res

    def forward(self, function_dist, target, *params, **kwargs):
        if not isinstance(function_dist, MultivariateNormal):
            raise RuntimeError("ExactMarginalLogLikelihood can only operate on Gaussian random variables")
        output = self.likelihood(function_dist, *params, **kwargs)
        res = output.log_prob(target)
        res = self._add_other_terms(res, params)
        num_data = function_dist.event_shape.numel()
        return res.div(num_data)
