from gpytorch_amd.operators import psd_safe_cholesky  # noqa: F401
