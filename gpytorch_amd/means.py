"""Mean functions used by the ExactGP hot path (``gpytorch/means/constant_mean.py:111-113``,
``zero_mean.py``).  Pure PyTorch glue."""
from __future__ import annotations

import torch

from .module import Module


class Mean(Module):
    def __call__(self, x):
        if x.dim() == 1:
            x = x.unsqueeze(1)
        return self.forward(x)


class ZeroMean(Mean):
    def forward(self, x):
        return torch.zeros(x.shape[:-1], dtype=x.dtype, device=x.device)


class ConstantMean(Mean):
    def __init__(self, constant_prior=None, constant_constraint=None, batch_shape=torch.Size()):
        super().__init__()
        self.batch_shape = torch.Size(batch_shape)
        self.register_parameter("raw_constant", torch.nn.Parameter(torch.zeros(self.batch_shape)))
        if constant_constraint is not None:
            self.register_constraint("raw_constant", constant_constraint)
        if constant_prior is not None:
            self.register_prior("mean_prior", constant_prior, self._constant_param, self._constant_closure)

    @staticmethod
    def _constant_param(m):        # constant_mean.py:99-101
        return m.constant

    @staticmethod
    def _constant_closure(m, value):
        m._set_transformed("raw_constant", value)

    @property
    def constant(self):
        return self._get_transformed("raw_constant")

    @constant.setter
    def constant(self, value):
        self._set_transformed("raw_constant", value)

    def forward(self, x):
        c = self.constant.unsqueeze(-1)  # constant_mean.py:111-113
        return c.expand(*torch.broadcast_shapes(c.shape[:-1], x.shape[:-2]), x.shape[-2])
