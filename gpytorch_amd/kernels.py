"""Kernel modules of the hot path: ``RBFKernel``, ``MaternKernel``, ``ScaleKernel`` (+ the ``Kernel``
base they share).  Same constructor arguments, parameter names (``raw_lengthscale``,
``raw_outputscale``), constraints and call semantics as ``gpytorch/kernels/kernel.py:84-589``,
``rbf_kernel.py``, ``matern_kernel.py``, ``scale_kernel.py`` -- but ``forward`` returns a matrix-free
:class:`~gpytorch_amd.operators.FusedKernelLinearOperator`, exactly like the in-tree precedent
``gpytorch/kernels/keops/rbf_kernel.py:44-55`` returns a ``KernelLinearOperator``.
"""
from __future__ import annotations

import math

import torch

from . import backend as B
from .functions import KernelSpec
from .module import Interval, Module, Positive
from .operators import BatchLinearOperator, FusedKernelLinearOperator, LinearOperator


class Kernel(Module):
    has_lengthscale = False

    def __init__(self, ard_num_dims=None, batch_shape=torch.Size([]), active_dims=None, lengthscale_prior=None,
                 lengthscale_constraint=None, eps=1e-6, **kwargs):
        super().__init__()
        self._batch_shape = torch.Size(batch_shape)
        if active_dims is not None and not torch.is_tensor(active_dims):
            active_dims = torch.tensor(active_dims, dtype=torch.long)
        self.register_buffer("active_dims", active_dims)
        self.ard_num_dims = ard_num_dims
        self.eps = eps
        if self.has_lengthscale:
            n_ls = 1 if ard_num_dims is None else ard_num_dims
            self.register_parameter("raw_lengthscale", torch.nn.Parameter(torch.zeros(*self._batch_shape, 1, n_ls)))
            self.register_constraint("raw_lengthscale", Positive() if lengthscale_constraint is None else lengthscale_constraint)
            if lengthscale_prior is not None:
                self.register_prior("lengthscale_prior", lengthscale_prior, lambda m: m.lengthscale, lambda m, v: m._set_lengthscale(v))

    @property
    def batch_shape(self):
        return self._batch_shape

    @property
    def lengthscale(self):
        return self._get_transformed("raw_lengthscale") if self.has_lengthscale else None

    @lengthscale.setter
    def lengthscale(self, value):
        self._set_lengthscale(value)

    def _set_lengthscale(self, value):
        if not self.has_lengthscale:
            raise RuntimeError("Kernel has no lengthscale.")
        self._set_transformed("raw_lengthscale", value)

    @property
    def is_stationary(self):
        return self.has_lengthscale

    def forward(self, x1, x2, diag=False, **params):
        raise NotImplementedError

    def __call__(self, x1, x2=None, diag=False, last_dim_is_batch=False, **params):
        """``kernel.py:454-534``: select active dims, promote 1-D inputs to [n, 1], default x2 = x1; leading dimensions of
        the inputs (and the kernel's ``batch_shape``) are batch dimensions (``kernel.py:163-208``)."""
        x1_, x2_ = x1, x2
        if last_dim_is_batch:  # kernel.py:506-510: every input dimension becomes its own batch member, [..., d, n, 1]
            x1_ = x1_.transpose(-1, -2).unsqueeze(-1)
            x2_ = None if x2_ is None else x2_.transpose(-1, -2).unsqueeze(-1)
        if x1_.dim() == 1:
            x1_ = x1_.unsqueeze(1)
        if x2_ is not None and x2_.dim() == 1:
            x2_ = x2_.unsqueeze(1)
        if self.active_dims is not None:
            x1_ = x1_.index_select(-1, self.active_dims)
            if x2_ is not None:
                x2_ = x2_.index_select(-1, self.active_dims)
        if x2_ is None:
            x2_ = x1_
        elif x1_.shape[-1] != x2_.shape[-1]:
            raise RuntimeError("x1_ and x2_ must have the same number of dimensions!")
        if self.ard_num_dims is not None and self.ard_num_dims != x1_.shape[-1]:
            raise RuntimeError(f"Expected the input to have {self.ard_num_dims} dimensionality (based on ard_num_dims). Got {x1_.shape[-1]}.")
        return self.forward(x1_, x2_, diag=diag, **params)

    @property
    def prediction_strategy(self):
        from .models import DefaultPredictionStrategy

        return DefaultPredictionStrategy


class _StationaryFused(Kernel):
    has_lengthscale = True
    kind = None

    def _shift(self, x1):
        # stationary kernel: any common shift is exact; centring keeps |z| small, which the Gram-form
        # generation kernel needs (the reference's sq_dist centres for the same reason, kernel.py:29-30)
        return x1.detach().mean(dim=-2)

    def forward(self, x1, x2, diag=False, **params):
        # float32 with d <= 16: fused MFMA / VALU kernels; float64 or d > 16: generic path (backend.kv_chunked)
        ls = self.lengthscale
        batch = torch.broadcast_shapes(x1.shape[:-2], x2.shape[:-2], ls.shape[:-2])
        if not batch:
            op = FusedKernelLinearOperator(x1, x2, KernelSpec(self.kind, self._shift(x1)), ls)
            return op.diagonal() if diag else op
        # batch mode: one fused operator per batch member (inputs and lengthscales broadcast against each other)
        same = x2 is x1
        x1b = x1.expand(*batch, *x1.shape[-2:]).reshape(-1, *x1.shape[-2:])
        x2b = x1b if same else x2.expand(*batch, *x2.shape[-2:]).reshape(-1, *x2.shape[-2:])
        lsb = ls.expand(*batch, *ls.shape[-2:]).reshape(-1, *ls.shape[-2:])
        ops = []
        for b in range(x1b.shape[0]):
            xa = x1b[b]
            xb = xa if same else x2b[b]
            ops.append(FusedKernelLinearOperator(xa, xb, KernelSpec(self.kind, self._shift(xa)), lsb[b]))
        op = BatchLinearOperator(ops, batch)
        return op.diagonal() if diag else op


class RBFKernel(_StationaryFused):
    r"""k(x, x') = exp(-1/2 (x - x')^T Theta^-2 (x - x'))  (``gpytorch/kernels/rbf_kernel.py:14-85``)."""

    kind = "rbf"


class MaternKernel(_StationaryFused):
    r"""Matern nu in {1/2, 3/2, 5/2} (``gpytorch/kernels/matern_kernel.py:14-110``); inputs are centred by
    the mean of x1 first, as the reference does (``matern_kernel.py:94-97``)."""

    def __init__(self, nu: float = 2.5, **kwargs):
        if nu not in {0.5, 1.5, 2.5}:
            raise RuntimeError("nu expected to be 0.5, 1.5, or 2.5")
        super().__init__(**kwargs)
        self.nu = nu

    @property
    def kind(self):
        return B.NU_TO_KIND[self.nu]

    def _shift(self, x1):
        return x1.detach().mean(dim=-2)


class ScaleKernel(Kernel):
    r"""K_scaled = outputscale * K_orig  (``gpytorch/kernels/scale_kernel.py:20-124``)."""

    def __init__(self, base_kernel, outputscale_prior=None, outputscale_constraint=None, **kwargs):
        if base_kernel.active_dims is not None:
            kwargs["active_dims"] = base_kernel.active_dims
        super().__init__(**kwargs)
        self.base_kernel = base_kernel
        self.register_parameter("raw_outputscale", torch.nn.Parameter(torch.zeros(self._batch_shape)))
        self.register_constraint("raw_outputscale", Positive() if outputscale_constraint is None else outputscale_constraint)
        if outputscale_prior is not None:
            self.register_prior("outputscale_prior", outputscale_prior, lambda m: m.outputscale, lambda m, v: m._set_outputscale(v))

    @property
    def is_stationary(self):
        return self.base_kernel.is_stationary

    @property
    def outputscale(self):
        return self._get_transformed("raw_outputscale")

    @outputscale.setter
    def outputscale(self, value):
        self._set_outputscale(value)

    def _set_outputscale(self, value):
        self._set_transformed("raw_outputscale", value)

    def __call__(self, x1, x2=None, diag=False, **params):
        # active_dims were inherited from the base kernel: let the base kernel apply them once
        return self.forward(x1, x2, diag=diag, **params)

    def forward(self, x1, x2, diag=False, **params):
        orig = self.base_kernel(x1, x2, diag=diag, **params)
        os_ = self.outputscale
        if diag:
            return orig * os_.unsqueeze(-1)
        if isinstance(orig, LinearOperator):  # scale_kernel.py:117-118: outputscale.view(*batch, 1, 1)
            return orig.mul(os_.reshape(1) if os_.numel() == 1 and not orig.batch_shape else os_)
        return orig * os_.reshape(*os_.shape, 1, 1)

    @property
    def prediction_strategy(self):
        return self.base_kernel.prediction_strategy


__all__ = ["Kernel", "RBFKernel", "MaternKernel", "ScaleKernel"]
_ = (math, Interval)
