"""Hyper-parameter priors: log p(theta) terms added to the marginal log-likelihood
(``gpytorch/mlls/exact_marginal_log_likelihood.py:41-52`` sums ``prior.log_prob(closure(module))`` over
``model.named_priors()``).  Same constructor arguments and ``transform=`` hook as
``gpytorch/priors/torch_priors.py`` / ``smoothed_box_prior.py``; each prior is a ``torch.distributions`` distribution that is
also an ``nn.Module`` whose parameters move with ``.to(device)``.
"""
from __future__ import annotations

import math

import torch
from torch import nn
from torch.distributions import Gamma, HalfCauchy, HalfNormal, LogNormal, Normal, Uniform
from torch.distributions import constraints as tconstraints


class Prior(nn.Module):
    """Mixin: ``log_prob`` applies the optional transform first (``priors/prior.py:26-34``); distribution parameters are
    registered as buffers so that ``.to(...)`` / ``state_dict`` see them."""

    _param_names: tuple = ()

    def _init_prior(self, transform):
        self._transform = transform
        for name in self._param_names:
            val = getattr(self, name)
            try:
                delattr(self, name)
            except AttributeError:
                pass
            self.register_buffer(name, torch.as_tensor(val, dtype=torch.get_default_dtype()).clone())

    def transform(self, x):
        return self._transform(x) if self._transform is not None else x

    def log_prob(self, x):
        return super().log_prob(self.transform(x))


def _make(dist_cls, names):
    class _P(Prior, dist_cls):
        _param_names = names

        def __init__(self, *args, validate_args=False, transform=None, **kwargs):
            nn.Module.__init__(self)
            dist_cls.__init__(self, *args, validate_args=validate_args, **kwargs)
            self._init_prior(transform)

        def expand(self, batch_shape):
            return type(self)(*[getattr(self, k).expand(torch.Size(batch_shape)) for k in names])

    return _P


class NormalPrior(_make(Normal, ("loc", "scale"))):
    """pdf(x) = N(x; loc, scale^2)  (``torch_priors.py:15-32``)."""


class LogNormalPrior(_make(LogNormal, ())):
    """``torch_priors.py:54-69``.  (LogNormal is a transformed distribution: its parameters live in ``base_dist``.)"""

    def __init__(self, loc, scale, validate_args=False, transform=None):
        nn.Module.__init__(self)
        LogNormal.__init__(self, torch.as_tensor(loc, dtype=torch.get_default_dtype()), torch.as_tensor(scale, dtype=torch.get_default_dtype()),
                           validate_args=validate_args)
        self._transform = transform

    def _apply(self, fn):
        self.base_dist.loc = fn(self.base_dist.loc)
        self.base_dist.scale = fn(self.base_dist.scale)
        return super()._apply(fn)

    def expand(self, batch_shape):
        return LogNormalPrior(self.loc.expand(torch.Size(batch_shape)), self.scale.expand(torch.Size(batch_shape)))


class GammaPrior(_make(Gamma, ("concentration", "rate"))):
    """pdf(x) = rate^conc / Gamma(conc) x^(conc-1) exp(-rate x)  (``torch_priors.py:105-123``)."""


class HalfNormalPrior(_make(HalfNormal, ())):
    def __init__(self, scale, validate_args=False, transform=None):
        nn.Module.__init__(self)
        HalfNormal.__init__(self, torch.as_tensor(scale, dtype=torch.get_default_dtype()), validate_args=validate_args)
        self._transform = transform

    def _apply(self, fn):
        self.base_dist.scale = fn(self.base_dist.scale)
        return super()._apply(fn)


class HalfCauchyPrior(_make(HalfCauchy, ())):
    def __init__(self, scale, validate_args=False, transform=None):
        nn.Module.__init__(self)
        HalfCauchy.__init__(self, torch.as_tensor(scale, dtype=torch.get_default_dtype()), validate_args=validate_args)
        self._transform = transform

    def _apply(self, fn):
        self.base_dist.scale = fn(self.base_dist.scale)
        return super()._apply(fn)


class UniformPrior(_make(Uniform, ("low", "high"))):
    """``torch_priors.py:72-84``."""


class SmoothedBoxPrior(Prior):
    """Box [a, b] with Gaussian tails of scale sigma outside it (``priors/smoothed_box_prior.py:14-106``):
    log p(x) = -log(b - a + sqrt(2 pi) sigma) - 0.5 * (max(|x - c| - r, 0) / sigma)^2,  c = (a+b)/2, r = (b-a)/2."""

    arg_constraints = {"sigma": tconstraints.positive, "a": tconstraints.real, "b": tconstraints.real}
    support = tconstraints.real

    def __init__(self, a, b, sigma=0.01, validate_args=False, transform=None):
        super().__init__()
        a, b, sigma = (torch.as_tensor(v, dtype=torch.get_default_dtype()).reshape(-1) for v in (a, b, sigma))
        if bool((b < a).any()):
            raise ValueError("must have that a < b (element-wise)")
        self.register_buffer("a", a.clone())
        self.register_buffer("b", b.clone())
        self.register_buffer("sigma", sigma.clone())
        self._transform = transform

    def log_prob(self, x):
        x = self.transform(x)
        c, r = (self.a + self.b) / 2, (self.b - self.a) / 2
        x = x.view(-1, self.a.shape[0]) if self.a.shape[0] > 1 else x
        dist = ((x - c).abs() - r).clamp(min=0)
        lp = -torch.log(2 * r + math.sqrt(2 * math.pi) * self.sigma) - 0.5 * (dist / self.sigma).pow(2)
        return lp.sum(-1) if self.a.shape[0] > 1 else lp
