"""``linear_operator.settings`` of the test shim: the setting classes of ``gpytorch_amd.settings`` (one shared state, so a context entered through
the reference's ``gpytorch.settings`` governs the device path)."""
import torch

from gpytorch_amd import settings as _s

for _n in ("cg_tolerance", "cholesky_jitter", "cholesky_max_tries", "ciq_samples", "deterministic_probes", "fast_computations", "max_cg_iterations",
           "max_cholesky_size", "max_lanczos_quadrature_iterations", "max_preconditioner_size", "max_root_decomposition_size",
           "min_preconditioning_size", "minres_tolerance", "num_contour_quadrature", "num_trace_samples", "preconditioner_tolerance",
           "skip_logdet_forward", "terminate_cg_by_size", "tridiagonal_jitter", "verbose_linalg"):
    globals()[_n] = getattr(_s, _n)


class use_toeplitz(_s._feature_flag):
    _default = True


class _linalg_dtype_cholesky(_s._dtype_value_context):
    _global_float_value = torch.double
    _global_double_value = torch.double


class _linalg_dtype_symeig(_s._dtype_value_context):
    _global_float_value = torch.double
    _global_double_value = torch.double


class linalg_dtypes:
    def __init__(self, default=torch.double, symeig=None, cholesky=None):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *args):
        return False
