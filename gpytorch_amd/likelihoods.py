"""Gaussian likelihoods: ``marginal()`` adds the noise diagonal to the latent covariance, producing the
``K + sigma^2 I`` operator the hot path solves with (``gpytorch/likelihoods/gaussian_likelihood.py:117-180``,
``noise_models.py:29-92``; noise constraint ``GreaterThan(1e-4)``)."""
from __future__ import annotations

import warnings

import torch

from . import settings
from .distributions import MultivariateNormal
from .linear_cg import NumericalWarning
from .module import AttrGetter, AttrSetter, GreaterThan, Module
from .operators import ConstantDiagLinearOperator, DiagLinearOperator, FixedPlusConstantDiagLinearOperator, ZeroLinearOperator


class GPInputWarning(UserWarning):
    """Same role as ``gpytorch.utils.warnings.GPInputWarning``."""


class HomoskedasticNoise(Module):
    def __init__(self, noise_prior=None, noise_constraint=None, batch_shape=torch.Size()):
        super().__init__()
        self.register_parameter("raw_noise", torch.nn.Parameter(torch.zeros(*batch_shape, 1)))
        self.register_constraint("raw_noise", GreaterThan(1e-4) if noise_constraint is None else noise_constraint)
        if noise_prior is not None:
            self.register_prior("noise_prior", noise_prior, AttrGetter("noise"), AttrSetter("_set_transformed", "raw_noise"))

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

    def get_fantasy_likelihood(self, **kwargs):
        """gaussian_likelihood.py:60-61 / likelihood.py: a homoskedastic likelihood serves the fantasy model unchanged (hyper-parameters shared)."""
        return self

    def __call__(self, input, *args, **kwargs):
        """likelihood.py:72-84: an MVN input means ``marginal``."""
        if isinstance(input, MultivariateNormal):
            return self.marginal(input, *args, **kwargs)
        raise RuntimeError("Likelihoods expects a MultivariateNormal input to make marginal predictions")


class GaussianLikelihood(_GaussianLikelihoodBase):
    def __init__(self, noise_prior=None, noise_constraint=None, batch_shape=torch.Size(), **kwargs):
        super().__init__(HomoskedasticNoise(noise_prior=noise_prior, noise_constraint=noise_constraint, batch_shape=batch_shape))

    @property
    def noise(self):
        return self.noise_covar.noise

    @noise.setter
    def noise(self, value):
        self.noise_covar.initialize(noise=value)

    @property
    def raw_noise(self):
        return self.noise_covar.raw_noise


class FixedGaussianNoise(Module):
    """Known per-point observation noise (``noise_models.py:138-176``): the stored diagonal when the requested shape
    matches it, a caller-supplied ``noise=`` diagonal when given, otherwise nothing (zero operator)."""

    def __init__(self, noise: torch.Tensor):
        super().__init__()
        floor = settings.min_fixed_noise.value(noise.dtype)
        if bool(noise.lt(floor).any()):
            warnings.warn(f"Very small noise values detected. This will likely lead to numerical instabilities. "
                          f"Rounding small noise values up to {floor}.", NumericalWarning)
            noise = noise.clamp_min(floor)
        self.register_buffer("noise", noise)

    def forward(self, *params, shape=None, noise=None, **kwargs):
        if shape is None:
            first = params[0] if torch.is_tensor(params[0]) else params[0][0]
            shape = first.shape if first.dim() == 1 else first.shape[:-1]
        diag = noise if noise is not None else (self.noise if shape[-1] == self.noise.shape[-1] else None)
        if diag is None:
            return ZeroLinearOperator(shape[-1], shape[-1], dtype=self.noise.dtype, device=self.noise.device)
        if diag.numel() > 0 and bool((diag == diag.reshape(-1)[0]).all()):
            return ConstantDiagLinearOperator(diag.reshape(-1)[:1], diag_shape=shape[-1])   # rides the constant-diagonal fused path
        return DiagLinearOperator(diag)


class FixedNoiseGaussianLikelihood(_GaussianLikelihoodBase):
    """``gaussian_likelihood.py:245-362``: fixed (heteroskedastic) training noise, optional learned extra noise,
    ``noise=`` for test-time noise.  The vector rides in the fused K*V epilogue (``FusedKernelAddedDiagLinearOperator``)."""

    def __init__(self, noise: torch.Tensor, learn_additional_noise: bool = False, batch_shape=torch.Size(), **kwargs):
        super().__init__(FixedGaussianNoise(noise))
        self.second_noise_covar = None
        if learn_additional_noise:
            # (gaussian_likelihood.py:291-296: the learned noise carries the likelihood's batch_shape -- one value per member of a batch of GPs)
            self.second_noise_covar = HomoskedasticNoise(noise_prior=kwargs.get("noise_prior"), noise_constraint=kwargs.get("noise_constraint"),
                                                         batch_shape=batch_shape)

    @property
    def noise(self):
        return self.noise_covar.noise + self.second_noise

    @noise.setter
    def noise(self, value):
        self.noise_covar.noise = torch.as_tensor(value).to(self.noise_covar.noise)

    @property
    def second_noise(self):
        return 0.0 if self.second_noise_covar is None else self.second_noise_covar.noise

    @second_noise.setter
    def second_noise(self, value):
        if self.second_noise_covar is None:
            raise RuntimeError("Attempting to set secondary learned noise for FixedNoiseGaussianLikelihood, "
                               "but learn_additional_noise must have been False!")
        self.second_noise_covar.initialize(noise=value)

    def get_fantasy_likelihood(self, **kwargs):
        """gaussian_likelihood.py:322-335: the fantasy points bring their own ``noise``."""
        if "noise" not in kwargs:
            raise RuntimeError("FixedNoiseGaussianLikelihood.fantasize requires a `noise` kwarg")
        import copy

        new = copy.deepcopy(self)
        new.noise_covar = FixedGaussianNoise(torch.cat([self.noise_covar.noise, kwargs["noise"].to(self.noise_covar.noise)], -1))
        return new

    def _shaped_noise_covar(self, base_shape, *params, **kwargs):
        shape = None if len(params) > 0 else base_shape
        res = self.noise_covar(*params, shape=shape, **kwargs)
        if self.second_noise_covar is not None:
            extra = self.second_noise_covar(*params, shape=base_shape)
            res = extra if isinstance(res, ZeroLinearOperator) else _add_diags(res, extra)
        elif isinstance(res, ZeroLinearOperator):
            warnings.warn("You have passed data through a FixedNoiseGaussianLikelihood that did not match the size of the "
                          "fixed noise, *and* you did not specify noise. This is treated as a no-op.", GPInputWarning)
        return res


class DirichletClassificationLikelihood(FixedNoiseGaussianLikelihood):
    """Classification labels as heteroskedastic regression targets (``gaussian_likelihood.py:365-472``; Milios et al., 2018): class c of point i gets
    the Dirichlet concentration alpha = alpha_epsilon (+ 1 for the observed class), matched by a log-normal with variance
    sigma2 = log(1 / alpha + 1) and mean log(alpha) - sigma2 / 2.  The result is ONE exact GP per class -- ``batch_shape = (classes,)`` -- with the
    fixed noise sigma2 [classes, n] and the targets ``transformed_targets`` [classes, n]: everything downstream is the fixed-noise hot path."""

    @staticmethod
    def _prepare_targets(targets, alpha_epsilon: float = 0.01, dtype=torch.float):
        classes = int(targets.max()) + 1
        alpha = torch.full((targets.shape[-1], classes), alpha_epsilon, device=targets.device, dtype=dtype)
        alpha[torch.arange(targets.shape[-1], device=targets.device), targets] += 1.0
        sigma2 = torch.log1p(alpha.reciprocal())
        return sigma2.mT.contiguous(), alpha.log() - 0.5 * sigma2, classes

    def __init__(self, targets, alpha_epsilon: float = 0.01, learn_additional_noise: bool = False, batch_shape=torch.Size(), dtype=torch.float, **kwargs):
        sigma2, transformed, classes = self._prepare_targets(targets, alpha_epsilon=alpha_epsilon, dtype=dtype)
        super().__init__(noise=sigma2, learn_additional_noise=learn_additional_noise, batch_shape=torch.Size((classes,)), **kwargs)
        self.transformed_targets = transformed.mT.contiguous()
        self.num_classes = classes
        self.targets = targets
        self.alpha_epsilon = alpha_epsilon

    def _apply(self, fn, *args, **kwargs):
        # (plain attributes in the reference too; moved with the module here so that ``likelihood.to(device)`` keeps the notebook's two lines working)
        self.transformed_targets = fn(self.transformed_targets)
        self.targets = fn(self.targets)
        return super()._apply(fn, *args, **kwargs)

    def get_fantasy_likelihood(self, **kwargs):
        """gaussian_likelihood.py:438-458: the fantasy points bring class labels (under the ``noise`` keyword, as the reference has it)."""
        if "noise" not in kwargs:
            raise RuntimeError("FixedNoiseGaussianLikelihood.fantasize requires a `noise` kwarg")
        import copy

        new = copy.deepcopy(self)
        labels = kwargs["noise"]
        sigma2, _, _ = self._prepare_targets(labels, self.alpha_epsilon, dtype=self.transformed_targets.dtype)
        old = self.noise_covar.noise
        if old.dim() != sigma2.dim():
            old = old.expand(*sigma2.shape[:-1], old.shape[-1])
        new.targets = torch.cat([self.targets, labels], -1)
        new.noise_covar = FixedGaussianNoise(torch.cat([old, sigma2.to(old)], -1))
        return new

    def __call__(self, input, *args, **kwargs):
        if "targets" in kwargs:      # test-time labels -> test-time noise (gaussian_likelihood.py:466-472)
            kwargs["noise"] = self._prepare_targets(kwargs.pop("targets"), dtype=self.transformed_targets.dtype)[0]
        return super().__call__(input, *args, **kwargs)


def _add_diags(a, b):
    """Sum of two diagonal operators as ONE diagonal operator (so the fused K + D path still applies)."""
    if isinstance(a, ConstantDiagLinearOperator) and isinstance(b, ConstantDiagLinearOperator):
        return ConstantDiagLinearOperator(a.diag_values.reshape(-1)[:1] + b.diag_values.reshape(-1)[:1], a.diag_shape)
    # fixed vector + learned scalar: kept apart so that the scalar stays on the autograd path of the fused operators
    # (batch mode: fixed [*batch, n] + a scalar that is shared ([1]) or per member ([*batch, 1]); BatchLinearOperator.__add__ splits both per member)
    if isinstance(b, ConstantDiagLinearOperator) and not isinstance(a, ConstantDiagLinearOperator):
        return FixedPlusConstantDiagLinearOperator(a._diag, b.diag_values)
    if isinstance(a, ConstantDiagLinearOperator) and not isinstance(b, ConstantDiagLinearOperator):
        return FixedPlusConstantDiagLinearOperator(b._diag, a.diag_values)
    return DiagLinearOperator(a._diag + b._diag)
