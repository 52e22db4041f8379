"""TEST-ONLY stand-in for the third-party ``linear_operator`` package (not installable here), so that the REFERENCE's own
Python layers (``/root/reference/gpytorch``: ExactGP, GaussianLikelihood, ExactMarginalLogLikelihood, MultivariateNormal,
DefaultPredictionStrategy) can be imported and executed, unmodified, over this repository's operators.

It holds no algorithm: every name the reference imports is mapped onto the class of the same role in ``gpytorch_amd.operators`` /
``gpytorch_amd.settings``; names off the exact-GP path (interpolation, Toeplitz, Kronecker-diag ...) are placeholder classes created on
demand that raise when instantiated.  Never imported by the product (``tests/test_lib_abi.py::test_product_never_imports_test_shim``).
"""
from gpytorch_amd import operators as _o

from . import operators, settings, utils  # noqa: F401
from .operators import LinearOperator, to_dense, to_linear_operator  # noqa: F401

__version__ = "0.6.shim"


def add_diagonal(input, diag):
    return to_linear_operator(input).add_diagonal(diag)


def add_jitter(input, jitter_val=1e-3):
    return to_linear_operator(input).add_jitter(jitter_val)


def inv_quad(input, inv_quad_rhs, reduce_inv_quad=True):
    return to_linear_operator(input).inv_quad(inv_quad_rhs, reduce_inv_quad=reduce_inv_quad)


def inv_quad_logdet(input, inv_quad_rhs=None, logdet=False, reduce_inv_quad=True):
    return to_linear_operator(input).inv_quad_logdet(inv_quad_rhs, logdet, reduce_inv_quad)


def solve(input, rhs, lhs=None):
    return to_linear_operator(input).solve(rhs, lhs)


def root_decomposition(input, method=None):
    return to_linear_operator(input).root_decomposition(method=method)


def root_inv_decomposition(input, initial_vectors=None, test_vectors=None, method=None):
    return to_linear_operator(input).root_inv_decomposition(initial_vectors, test_vectors, method)


def pivoted_cholesky(input, rank, error_tol=None, return_pivots=False):
    return to_linear_operator(input).pivoted_cholesky(rank, error_tol, return_pivots)


def diagonalization(input, method=None):
    raise NotImplementedError("linear_operator shim: diagonalization is off the exact-GP path")


def dsmm(sparse_mat, dense_mat):
    raise NotImplementedError("linear_operator shim: dsmm is off the exact-GP path")


def sqrt_inv_matmul(input, rhs, lhs=None):
    raise NotImplementedError("linear_operator shim: sqrt_inv_matmul is off the exact-GP path")
