"""Global flags / values of the BBMM path, mirroring ``gpytorch/settings.py`` (which re-exports the
``linear_operator.settings`` knobs at ``settings.py:6-31``).  Same names, same defaults, same
context-manager usage::

    with gpytorch_amd.settings.cg_tolerance(1e-4), gpytorch_amd.settings.max_cholesky_size(0):
        ...

Process-global class attributes exactly like the reference (``settings.py:84-144``): not thread-safe.
"""
from __future__ import annotations

import torch


class _feature_flag:
    """``settings.py:84-119``."""

    _default = False
    _state = None

    @classmethod
    def is_default(cls):
        return cls._state is None

    @classmethod
    def on(cls):
        return cls._default if cls.is_default() else cls._state

    @classmethod
    def off(cls):
        return not cls.on()

    @classmethod
    def _set_state(cls, state):
        cls._state = state

    def __init__(self, state=True):
        self.prev = self.__class__._state
        self.state = state

    def __enter__(self):
        self.__class__._set_state(self.state)

    def __exit__(self, *args):
        self.__class__._set_state(self.prev)
        return False


class _value_context:
    """``settings.py:122-144``."""

    _global_value = None

    @classmethod
    def value(cls):
        return cls._global_value

    @classmethod
    def _set_value(cls, value):
        cls._global_value = value

    def __init__(self, value):
        self._orig_value = self.__class__.value()
        self._instance_value = value

    def __enter__(self):
        self.__class__._set_value(self._instance_value)

    def __exit__(self, *args):
        self.__class__._set_value(self._orig_value)
        return False


class _dtype_value_context:
    """``settings.py:35-81``: per-dtype values (float / double / half)."""

    _global_float_value = None
    _global_double_value = None
    _global_half_value = None

    @classmethod
    def value(cls, dtype):
        if torch.is_tensor(dtype):
            dtype = dtype.dtype
        if dtype == torch.float:
            return cls._global_float_value
        if dtype == torch.double:
            return cls._global_double_value
        if dtype == torch.half:
            return cls._global_half_value
        raise RuntimeError(f"Unsupported dtype for {cls.__name__}.")

    def __init__(self, float_value=None, double_value=None, half_value=None):
        self._new = (float_value, double_value, half_value)
        c = self.__class__
        self._orig = (c._global_float_value, c._global_double_value, c._global_half_value)

    def __enter__(self):
        c = self.__class__
        f, d, h = self._new
        if f is not None:
            c._global_float_value = f
        if d is not None:
            c._global_double_value = d
        if h is not None:
            c._global_half_value = h

    def __exit__(self, *args):
        c = self.__class__
        c._global_float_value, c._global_double_value, c._global_half_value = self._orig
        return False


# ---- BBMM knobs (defaults of linear_operator.settings v0.6.x) ----
class cg_tolerance(_value_context):
    """Relative residual tolerance of mBCG during training (default 1)."""
    _global_value = 1.0


class eval_cg_tolerance(_value_context):
    """``gpytorch/settings.py:173-180``: CG tolerance used for predictions (default 0.01)."""
    _global_value = 1e-2


class max_cg_iterations(_value_context):
    _global_value = 1000


class max_cholesky_size(_value_context):
    """Below this size the reference uses dense Cholesky instead of CG (default 800)."""
    _global_value = 800


class max_lanczos_quadrature_iterations(_value_context):
    _global_value = 20


class max_preconditioner_size(_value_context):
    _global_value = 15


class min_preconditioning_size(_value_context):
    _global_value = 2000


class preconditioner_tolerance(_value_context):
    _global_value = 1e-3


class max_root_decomposition_size(_value_context):
    _global_value = 100


class num_trace_samples(_value_context):
    _global_value = 10


class tridiagonal_jitter(_value_context):
    _global_value = 1e-6


class terminate_cg_by_size(_feature_flag):
    _default = False


class skip_logdet_forward(_feature_flag):
    _default = False


class skip_posterior_variances(_feature_flag):
    """``gpytorch/settings.py:360-370``."""
    _default = False


class fast_pred_var(_feature_flag):
    """``gpytorch/settings.py:183-222``: LOVE predictive variances."""
    _default = False


class detach_test_caches(_feature_flag):
    _default = True


class lazily_evaluate_kernels(_feature_flag):
    """``gpytorch/settings.py:246-258``."""
    _default = True


class verbose_linalg(_feature_flag):
    _default = False


class debug(_feature_flag):
    _default = True


class max_eager_kernel_size(_value_context):
    """``gpytorch/settings.py:261-269``."""
    _global_value = 512


class min_variance(_dtype_value_context):
    """``gpytorch/settings.py:299-311``."""
    _global_float_value = 1e-6
    _global_double_value = 1e-10
    _global_half_value = 1e-3


class cholesky_jitter(_dtype_value_context):
    _global_float_value = 1e-6
    _global_double_value = 1e-8
    _global_half_value = 1e-4


class deterministic_probes(_feature_flag):
    """Re-use one fixed set of probe vectors (``linear_operator.settings.deterministic_probes``).
    ``deterministic_probes.probe_vectors`` may be pre-set to an (n, t) tensor to inject Z."""
    _default = False
    probe_vectors = None


class fast_computations:
    """``linear_operator.settings.fast_computations``: three independent flags."""

    class covar_root_decomposition(_feature_flag):
        _default = True

    class log_prob(_feature_flag):
        _default = True

    class solves(_feature_flag):
        _default = True

    def __init__(self, covar_root_decomposition=True, log_prob=True, solves=True):
        self._ctx = [
            fast_computations.covar_root_decomposition(covar_root_decomposition),
            fast_computations.log_prob(log_prob),
            fast_computations.solves(solves),
        ]

    def __enter__(self):
        for c in self._ctx:
            c.__enter__()

    def __exit__(self, *args):
        for c in self._ctx:
            c.__exit__()
        return False


class sharding:
    """Process-group defaults for the multi-GPU paths (one process per GPU; ``torchrun``), read by the fused operators
    when their ``bbmm_opts`` do not say otherwise:

      * ``probe_group``: the probe columns of ``inv_quad_logdet`` (MLL forward / backward) are partitioned over the ranks
        of this group -- each rank draws ``num_trace_samples // world`` (+1 for the first ranks) probes;
      * ``row_group``: the small-t solves of the predictive posterior (mean-cache CG, LOVE Lanczos) are ROW-sharded over
        this group -- every rank ends up with the full caches.

    Replaces ``gpytorch.kernels.MultiDeviceKernel`` (``multi_device_kernel.py:49-92``).  Not thread-safe, like every
    other setting here (``gpytorch/settings.py:84-144`` are process-global class attributes)."""

    _probe_group = None
    _row_group = None

    def __init__(self, probe_group=None, row_group=None):
        self._new = (probe_group, row_group)

    @classmethod
    def probe_group(cls):
        return cls._probe_group

    @classmethod
    def row_group(cls):
        return cls._row_group

    def __enter__(self):
        self._old = (sharding._probe_group, sharding._row_group)
        sharding._probe_group, sharding._row_group = self._new
        return self

    def __exit__(self, *args):
        sharding._probe_group, sharding._row_group = self._old
        return False
