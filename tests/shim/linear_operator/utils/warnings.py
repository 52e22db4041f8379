from gpytorch_amd.linear_cg import NumericalWarning  # noqa: F401
