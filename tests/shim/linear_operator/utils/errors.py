from gpytorch_amd.operators import NanError, NotPSDError  # noqa: F401


class CachingError(RuntimeError):
    pass
