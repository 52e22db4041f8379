"""Dual-mode boundary: the fused path as a plugin of the REAL ``gpytorch`` / ``linear_operator`` when they are importable.

The standalone classes of this package (``gpytorch_amd.kernels`` / ``operators`` / ``mlls`` ...) mirror the reference's API because
neither ``gpytorch`` nor ``linear_operator`` can be imported in the build container.  Where they CAN be imported, the
reference's own extension seam is used instead -- exactly what ``gpytorch/kernels/keops/rbf_kernel.py:18-55`` does for KeOps:

  * ``RBFKernel`` / ``MaternKernel`` subclass ``gpytorch.kernels.Kernel``; ``forward`` returns
  * ``FusedKernelOperator``, a ``linear_operator.LinearOperator`` whose ``_matmul`` / ``_diagonal`` / ``_getitem`` /
    ``_bilinear_derivative`` run on ``libgpamd.so`` and whose ``+ noise`` yields
  * ``FusedAddedDiagOperator``, which overrides ``inv_quad_logdet`` / ``solve`` / ``root_inv_decomposition`` /
    ``_preconditioner`` with the device-resident BBMM loop;
  * ``ExactMarginalLogLikelihood``, ``ExactGP``, the likelihoods, means and ``MultivariateNormal`` are the reference's own,
    unchanged (``dropin.ExactMarginalLogLikelihood is gpytorch.mlls.ExactMarginalLogLikelihood``).

``AVAILABLE`` tells whether the real packages were found; ``build(gpytorch_module, linear_operator_module)`` constructs the classes
against any pair of modules with that interface (used by the tests to check the wiring against minimal stand-ins).
"""
from __future__ import annotations

import types

import torch

from . import backend as B
from . import operators as _ops
from .functions import KernelSpec, hyper_grads

try:  # pragma: no cover - depends on the environment
    import gpytorch as _gpytorch
    import linear_operator as _linear_operator

    AVAILABLE = True
except Exception:  # noqa: BLE001 - any import problem means "standalone mode"
    _gpytorch = _linear_operator = None
    AVAILABLE = False


def build(gp, lo) -> types.SimpleNamespace:
    """Create the plugin classes against ``gp`` (gpytorch-like) and ``lo`` (linear_operator-like)."""
    LinearOperator = lo.operators.LinearOperator

    class FusedKernelOperator(LinearOperator):
        """outputscale * k(x1, x2), never formed.  Tensor arguments go to ``super().__init__`` so that ``representation()`` /
        ``representation_tree()`` can rebuild the operator inside linear_operator's autograd Functions."""

        def __init__(self, x1, x2, lengthscale, outputscale=None, kind="rbf", shift=None):
            super().__init__(x1, x2, lengthscale, outputscale, kind=kind, shift=shift)
            self.x1, self.x2, self.lengthscale, self.outputscale, self.kind, self.shift = x1, x2, lengthscale, outputscale, kind, shift
            self._inner = _ops.FusedKernelLinearOperator(x1, x2, KernelSpec(kind, shift), lengthscale, outputscale)

        def _size(self):
            return self._inner._size()

        def _transpose_nonbatch(self):
            return FusedKernelOperator(self.x2, self.x1, self.lengthscale, self.outputscale, self.kind, self.shift)

        def _matmul(self, rhs):
            return self._inner._matmul(rhs)

        def _diagonal(self):
            return self._inner.diagonal()

        def to_dense(self):
            return self._inner.to_dense()

        def _getitem(self, row_index, col_index, *batch_indices):
            sub = self._inner[row_index, col_index]
            if isinstance(sub, torch.Tensor):
                return lo.to_linear_operator(sub)
            return FusedKernelOperator(sub.x1, sub.x2, self.lengthscale, self.outputscale, self.kind, self.shift)

        def _mul_constant(self, other):
            os_ = other if self.outputscale is None else self.outputscale * other
            return FusedKernelOperator(self.x1, self.x2, self.lengthscale, os_.reshape(1), self.kind, self.shift)

        def _bilinear_derivative(self, left_vecs, right_vecs):
            """Gradients of sum_c left_c^T K right_c for every tensor of ``representation()`` = (x1, x2, lengthscale, outputscale)."""
            p1, p2 = self._inner.prepared()
            lt, rt = B.to_probe_major(left_vecs, p1.dtype), B.to_probe_major(right_vecs, p1.dtype)
            want_x = self.x1.requires_grad or self.x2.requires_grad
            if want_x:
                d_ls, d_os, gx1, gx2 = hyper_grads(p1, p2, self.lengthscale, self.outputscale, lt, rt, want_x1=True, want_x2=True)
            else:
                d_ls, d_os = hyper_grads(p1, p2, self.lengthscale, self.outputscale, lt, rt)
                gx1 = gx2 = None
            grads = (gx1, gx2, d_ls) + (() if self.outputscale is None else (d_os,))
            return tuple(grads)

        def add_diagonal(self, diag):
            if diag.numel() == 1 and self.is_square:
                return FusedAddedDiagOperator(self, diag.reshape(1))
            return super().add_diagonal(diag)

        def __add__(self, other):
            if isinstance(other, lo.operators.ConstantDiagLinearOperator) and self.is_square and other.diag_values.numel() == 1:
                return FusedAddedDiagOperator(self, other.diag_values.reshape(1))
            return super().__add__(other)

    class FusedAddedDiagOperator(LinearOperator):
        """K + sigma^2 I with the whole BBMM loop on the device (mBCG, pivoted-Cholesky preconditioner, SLQ, Lanczos)."""

        def __init__(self, kernel_op, noise):
            super().__init__(kernel_op, noise)
            self.kernel_op, self.noise = kernel_op, noise
            self._inner = _ops.FusedKernelAddedDiagLinearOperator(kernel_op._inner, noise)

        def _size(self):
            return self._inner._size()

        def _transpose_nonbatch(self):
            return self

        def _matmul(self, rhs):
            return self._inner._matmul(rhs)

        def _diagonal(self):
            return self._inner.diagonal()

        def to_dense(self):
            return self._inner.to_dense()

        def inv_quad_logdet(self, inv_quad_rhs=None, logdet=False, reduce_inv_quad=True):
            return self._inner.inv_quad_logdet(inv_quad_rhs, logdet, reduce_inv_quad)

        def solve(self, right_tensor, left_tensor=None):
            return self._inner.solve(right_tensor, left_tensor)

        def _preconditioner(self):
            return self._inner._preconditioner()

        def root_inv_decomposition(self, initial_vectors=None, test_vectors=None, method=None):
            root = self._inner.root_inv_decomposition(initial_vectors, test_vectors, method).root
            return lo.operators.RootLinearOperator(root)

    class _FusedStationary(gp.kernels.Kernel):
        has_lengthscale = True
        kind = "rbf"

        def forward(self, x1, x2, diag=False, **params):
            op = FusedKernelOperator(x1, x2, self.lengthscale, None, self.kind, x1.detach().mean(dim=-2))
            return op._diagonal() if diag else op

    class RBFKernel(_FusedStationary):
        """Drop-in for ``gpytorch.kernels.RBFKernel`` / ``gpytorch.kernels.keops.RBFKernel``."""

        kind = "rbf"

    class MaternKernel(_FusedStationary):
        """Drop-in for ``gpytorch.kernels.MaternKernel`` (nu in {1/2, 3/2, 5/2})."""

        def __init__(self, nu: float = 2.5, **kwargs):
            if nu not in {0.5, 1.5, 2.5}:
                raise RuntimeError("nu expected to be 0.5, 1.5, or 2.5")
            super().__init__(**kwargs)
            self.nu = nu

        @property
        def kind(self):
            return B.NU_TO_KIND[self.nu]

    return types.SimpleNamespace(
        FusedKernelOperator=FusedKernelOperator, FusedAddedDiagOperator=FusedAddedDiagOperator, RBFKernel=RBFKernel,
        MaternKernel=MaternKernel, ExactMarginalLogLikelihood=gp.mlls.ExactMarginalLogLikelihood,
    )


if AVAILABLE:  # pragma: no cover - depends on the environment
    _ns = build(_gpytorch, _linear_operator)
    FusedKernelOperator, FusedAddedDiagOperator = _ns.FusedKernelOperator, _ns.FusedAddedDiagOperator
    RBFKernel, MaternKernel = _ns.RBFKernel, _ns.MaternKernel
    ExactMarginalLogLikelihood = _ns.ExactMarginalLogLikelihood
