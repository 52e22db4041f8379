"""Gaussian likelihoods: ``marginal()`` adds the noise diagonal to the latent covariance, producing the
``K + sigma^2 I`` operator the hot path solves with (``gpytorch/likelihoods/gaussian_likelihood.py:117-180``,
``noise_models.py:29-92``; noise constraint ``GreaterThan(1e-4)``)."""
from __future__ import annotations

import torch

from .distributions import MultivariateNormal
from .module import GreaterThan, Module
from .operators import ConstantDiagLinearOperator, DiagLinearOperator


class HomoskedasticNoise(Module):
    def __init__(self, noise_prior=None, noise_constraint=None):
        super().__init__()
        self.register_parameter("raw_noise", torch.nn.Parameter(torch.zeros(1)))
        self.register_constraint("raw_noise", GreaterThan(1e-4) if noise_constraint is None else noise_constraint)
        if noise_prior is not None:
            self.register_prior("noise_prior", noise_prior, lambda m: m.noise, lambda m, v: m._set_transformed("raw_noise", v))

    @property
    def noise(self):
        return self._get_transformed("raw_noise")

    @noise.setter
    def noise(self, value):
        self._set_transformed("raw_noise", value)

    def forward(self, *params, shape=None, **kwargs):
        n = shape[-1]
        return ConstantDiagLinearOperator(self.noise, diag_shape=n)  # noise_models.py:92


class _GaussianLikelihoodBase(Module):
    def __init__(self, noise_covar):
        super().__init__()
        self.noise_covar = noise_covar

    def _shaped_noise_covar(self, base_shape, *params, **kwargs):
        return self.noise_covar(*params, shape=base_shape, **kwargs)

    def marginal(self, function_dist: MultivariateNormal, *params, **kwargs) -> MultivariateNormal:
        """gaussian_likelihood.py:117-121."""
        mean, covar = function_dist.mean, function_dist.lazy_covariance_matrix
        noise_covar = self._shaped_noise_covar(mean.shape, *params, **kwargs)
        return function_dist.__class__(mean, covar + noise_covar)

    def __call__(self, input, *args, **kwargs):
        """likelihood.py:72-84: an MVN input means ``marginal``."""
        if isinstance(input, MultivariateNormal):
            return self.marginal(input, *args, **kwargs)
        raise RuntimeError("Likelihoods expects a MultivariateNormal input to make marginal predictions")


class GaussianLikelihood(_GaussianLikelihoodBase):
    def __init__(self, noise_prior=None, noise_constraint=None, batch_shape=torch.Size(), **kwargs):
        super().__init__(HomoskedasticNoise(noise_prior=noise_prior, noise_constraint=noise_constraint))

    @property
    def noise(self):
        return self.noise_covar.noise

    @noise.setter
    def noise(self, value):
        self.noise_covar.initialize(noise=value)

    @property
    def raw_noise(self):
        return self.noise_covar.raw_noise


class FixedNoiseGaussianLikelihood(_GaussianLikelihoodBase):
    """gaussian_likelihood.py:245-362, constant-noise case rides the fused path; a heteroskedastic
    noise vector falls back to the generic (dense) operator sum."""

    class _Fixed(Module):
        def __init__(self, noise):
            super().__init__()
            self.register_buffer("noise", noise)

        def forward(self, *params, shape=None, **kwargs):
            if bool((self.noise == self.noise[0]).all()):
                return ConstantDiagLinearOperator(self.noise[:1], diag_shape=shape[-1])
            return DiagLinearOperator(self.noise)

    def __init__(self, noise: torch.Tensor, **kwargs):
        super().__init__(self._Fixed(noise))

    @property
    def noise(self):
        return self.noise_covar.noise
