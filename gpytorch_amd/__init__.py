"""gpytorch_amd -- MI355X-native BBMM inference path for exact Gaussian processes.

Drop-in (same names / argument meaning) for the hot path of cornellius-gp/gpytorch:
kernel-matrix MVM, modified batched CG, Lanczos / SLQ, pivoted-Cholesky preconditioner, behind
``kernels`` / ``LinearOperator``-protocol operators / ``ExactMarginalLogLikelihood``::

    import gpytorch_amd as gpytorch
    class GP(gpytorch.models.ExactGP):
        def __init__(self, x, y, lik):
            super().__init__(x, y, lik)
            self.mean_module = gpytorch.means.ConstantMean()
            self.covar_module = gpytorch.kernels.ScaleKernel(gpytorch.kernels.RBFKernel())
        def forward(self, x):
            return gpytorch.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

The compute path is hand-written HIP for gfx950 (``csrc/``, C ABI in ``include/gpamd.h``);
there is no CPU fallback.  The functional aliases below mirror ``gpytorch/__init__.py:34-278``.
"""
from . import constraints, distributed, distributions, kernels, likelihoods, means, mlls, models, operators, priors, settings  # noqa: F401
from ._lib import LIB_PATH, GpamdError  # noqa: F401
from .mlls import ExactMarginalLogLikelihood  # noqa: F401
from .module import Module  # noqa: F401
from .operators import to_dense, to_linear_operator
from . import multitask as _mt  # noqa: E402

# multitask members live where the reference keeps them
kernels.IndexKernel, kernels.MultitaskKernel = _mt.IndexKernel, _mt.MultitaskKernel
means.MultitaskMean = _mt.MultitaskMean
distributions.MultitaskMultivariateNormal = _mt.MultitaskMultivariateNormal
likelihoods.MultitaskGaussianLikelihood = _mt.MultitaskGaussianLikelihood

__version__ = "0.1.0"


def add_diagonal(input, diag):
    return to_linear_operator(input).add_diagonal(diag)


def add_jitter(input, jitter_val=1e-3):
    return to_linear_operator(input).add_jitter(jitter_val)


def inv_quad(input, inv_quad_rhs, reduce_inv_quad=True):
    return to_linear_operator(input).inv_quad(inv_quad_rhs, reduce_inv_quad=reduce_inv_quad)


def inv_quad_logdet(input, inv_quad_rhs=None, logdet=False, reduce_inv_quad=True):
    return to_linear_operator(input).inv_quad_logdet(inv_quad_rhs, logdet, reduce_inv_quad=reduce_inv_quad)


def logdet(input):
    return to_linear_operator(input).logdet()


def pivoted_cholesky(input, rank, error_tol=None, return_pivots=False):
    """gpytorch/__init__.py:146-173 (note: the reference wrapper drops error_tol at :173; honoured here)."""
    return to_linear_operator(input).pivoted_cholesky(rank=rank, error_tol=error_tol, return_pivots=return_pivots)


def root_decomposition(input):
    return to_linear_operator(input).root_decomposition()


def root_inv_decomposition(input, initial_vectors=None, test_vectors=None):
    return to_linear_operator(input).root_inv_decomposition(initial_vectors, test_vectors)


def solve(input, rhs, lhs=None):
    return to_linear_operator(input).solve(rhs, lhs)


def matmul(mat, rhs):
    """Deprecated alias kept by the reference (``gpytorch/functions/__init__.py:31-33``)."""
    import warnings

    warnings.warn("gpytorch.matmul is deprecated. Use torch.matmul instead.", DeprecationWarning)
    return mat @ rhs


def inv_matmul(mat, right_tensor, left_tensor=None):
    """Deprecated alias kept by the reference (``gpytorch/functions/__init__.py:36-38``): ``solve`` under its old name."""
    import warnings

    warnings.warn("gpytorch.inv_matmul is deprecated. Use gpytorch.solve instead.", DeprecationWarning)
    return solve(mat, right_tensor, left_tensor)


def sqrt_inv_matmul(input, rhs, lhs=None):
    """gpytorch/__init__.py:252-278: K^{-1/2} rhs by contour-integral quadrature + msMINRES (:mod:`gpytorch_amd.ciq`)."""
    from .ciq import sqrt_inv_matmul as _f

    return _f(to_linear_operator(input), rhs, lhs)


__all__ = [
    "ExactMarginalLogLikelihood", "Module", "add_diagonal", "add_jitter", "distributed", "distributions", "inv_matmul", "inv_quad", "inv_quad_logdet",
    "kernels", "likelihoods", "logdet", "matmul", "means", "mlls", "models", "operators", "pivoted_cholesky", "priors", "root_decomposition",
    "root_inv_decomposition", "settings", "solve", "sqrt_inv_matmul", "to_dense",
]
