"""gpytorch_amd -- MI355X-native BBMM inference path for exact Gaussian processes.

Drop-in (same names / argument meaning) for the hot path of cornellius-gp/gpytorch:
kernel-matrix MVM, modified batched CG, Lanczos / SLQ, pivoted-Cholesky preconditioner, behind
``kernels`` / ``LinearOperator``-protocol operators / ``ExactMarginalLogLikelihood``.
The compute path is hand-written HIP for gfx950 (``csrc/``, C ABI in ``include/gpamd.h``);
there is no CPU fallback.
"""
from . import settings  # noqa: F401
from ._lib import LIB_PATH, GpamdError  # noqa: F401

__version__ = "0.1.0"
