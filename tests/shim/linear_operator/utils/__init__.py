"""``linear_operator.utils`` of the test shim."""
from . import cholesky, errors, getitem, interpolation, warnings  # noqa: F401


def linear_cg(*args, **kwargs):
    raise NotImplementedError("linear_operator shim: the reference's exact-GP layers never call linear_cg directly (it is reached through "
                              "LinearOperator.solve / inv_quad_logdet, which the fused operators override)")
