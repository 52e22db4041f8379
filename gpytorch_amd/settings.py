"""Global flags / values of the BBMM path, mirroring ``gpytorch/settings.py`` (which re-exports the
``linear_operator.settings`` knobs at ``settings.py:6-31``).  Same names, same defaults, same
context-manager usage::

    with gpytorch_amd.settings.cg_tolerance(1e-4), gpytorch_amd.settings.max_cholesky_size(0):
        ...

Process-global class attributes exactly like the reference (``settings.py:84-144``): not thread-safe.
"""
from __future__ import annotations

import torch


_KEEP = object()


class _Scoped:
    """One process-global knob.  The state lives in the class attributes named by ``_fields``; an instance is a
    ``with`` scope that installs new values on entry and puts the previous ones back on exit (saved at ENTRY, so
    scopes nest and instances can be re-used).  Same surface as the three context bases of ``gpytorch/settings.py``
    (``_feature_flag.on()/off()``, ``_value_context.value()``, ``_dtype_value_context.value(dtype)``)."""

    _fields: tuple = ()

    def __init__(self, *new):
        self._new = new

    def __enter__(self):
        cls = type(self)
        self._saved = tuple(getattr(cls, f) for f in cls._fields)
        for f, v in zip(cls._fields, self._new):
            if v is not _KEEP:
                setattr(cls, f, v)
        return self

    def __exit__(self, *exc):
        cls = type(self)
        for f, v in zip(cls._fields, self._saved):
            setattr(cls, f, v)
        return False


class _feature_flag(_Scoped):
    """Boolean knob: ``_default`` unless a scope (or ``_set_state``) overrides it."""

    _fields = ("_state",)
    _default = False
    _state = None

    def __init__(self, state=True):
        super().__init__(bool(state))

    @classmethod
    def is_default(cls):
        return cls._state is None

    @classmethod
    def on(cls):
        return cls._default if cls._state is None else cls._state

    @classmethod
    def off(cls):
        return not cls.on()

    @classmethod
    def _set_state(cls, state):
        cls._state = state


class _value_context(_Scoped):
    """Scalar knob; the class attribute ``_global_value`` is the default."""

    _fields = ("_global_value",)
    _global_value = None

    def __init__(self, value):
        super().__init__(value)

    @classmethod
    def value(cls):
        return cls._global_value

    @classmethod
    def _set_value(cls, value):
        cls._global_value = value


class _dtype_value_context(_Scoped):
    """Knob with one value per floating dtype (float / double / half); ``None`` arguments keep the current value."""

    _fields = ("_global_float_value", "_global_double_value", "_global_half_value")
    _global_float_value = _global_double_value = _global_half_value = None

    def __init__(self, float_value=None, double_value=None, half_value=None):
        super().__init__(*(_KEEP if v is None else v for v in (float_value, double_value, half_value)))

    @classmethod
    def value(cls, dtype):
        dtype = dtype.dtype if torch.is_tensor(dtype) else dtype
        slot = {torch.float32: "_global_float_value", torch.float64: "_global_double_value", torch.float16: "_global_half_value"}.get(dtype)
        if slot is None:
            raise RuntimeError(f"Unsupported dtype for {cls.__name__}.")
        return getattr(cls, slot)


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
    """Rank of the pivoted-Cholesky preconditioner (``linear_operator.settings.max_preconditioner_size``, re-exported at
    ``gpytorch/settings.py:6-31``; default 15 = the reference's).  ``"auto"`` (no counterpart in the reference) picks the rank from the
    number of points: on MI355X the k pivot steps cost O(n k^2) memory traffic -- 17 ms at n = 500 000, k = 100; 57 ms at k = 256 -- against
    O(n^2 t) per mBCG iteration (91 ms there for 65 columns, 18.6 ms for one), and the iteration count falls steeply with the rank.  Metric shape
    (n = 500 000, RBF d = 3; ``profiles/r06_s9_*``): MLL at ``cg_tolerance`` 1, rank 0 / 15 / 100 / 128: 94 / 42 / 21 / 21 iterations (21 = the floor
    ``max_lanczos_quadrature_iterations`` sets), 8.6 / 4.0 / 2.08 / 2.10 s forward + backward, y^T K^-1 y within 3e-2 / 1.6e-1 / 1.3e-3 / 9e-5 of the
    converged value; mean-cache solve at ``eval_cg_tolerance`` 1e-4, rank 100 / 128 / 192 / 256 / 384: 121 / 70 / 28 / 13 / 11 iterations, cold
    posterior 3.1 / 2.1 / 1.35 / 1.29 / 1.33 s.  The rule keeps the build below a few products of the solve it serves: n >= 200 000 -> 256,
    n >= 40 000 -> 128, n >= 6 000 -> 50, else the reference's 15; the ``preconditioner_tolerance`` early stop applies as always, so a fast-decaying
    spectrum ends the factor earlier.  (Ranks up to 512 run on the fused kernels since round 6.)"""
    _global_value = 15
    auto_ranks = ((200_000, 256), (40_000, 128), (6_000, 50), (0, 15))

    @classmethod
    def resolve(cls, n: int) -> int:
        """The rank in force for a system of ``n`` points (the integer setting itself, or the ``"auto"`` rule above)."""
        v = cls.value()
        if v == "auto":
            for floor, rank in cls.auto_ranks:
                if n >= floor:
                    return rank
        return int(v)


class min_preconditioning_size(_value_context):
    _global_value = 2000


class preconditioner_tolerance(_value_context):
    _global_value = 1e-3


class max_root_decomposition_size(_value_context):
    """Rank of the Lanczos root decompositions (LOVE cache; ``linear_operator.settings.max_root_decomposition_size``, default 100).  With a block
    recurrence (``lanczos_block_size`` b > 1) the rank is ``b * (rank // b)`` -- 100 -> 96 at b = 8 --, and a block that turns out rank deficient
    in float32 ends the decomposition early with a ``NumericalWarning`` naming the rank reached (``lanczos.block_lanczos_steps``)."""
    _global_value = 100


class lanczos_block_size(_value_context):
    """(no counterpart in the reference.)  Rows per operator product of the Lanczos decompositions behind the LOVE cache
    (``root_inv_decomposition`` without start vectors): 1 = the reference's single-vector recurrence (``max_root_decomposition_size``
    one-column products, exp-bound on MI355X: 19 ms each at n = 500 000); b > 1 = block Lanczos (``lanczos.block_lanczos_steps``):
    ``max_root_decomposition_size // b`` products of b columns on the matrix-pipe kernels for a cache of the same rank -- the variance error
    of a LOVE cache follows its rank, not the way its Krylov space was generated (``tests/test_block_lanczos_cpu.py``).
    "auto" (default): 8 from ``auto_min_size`` rows on (a product then fills the chip) when the rank is at least ``auto_min_rank`` (200: below it a
    block cache is measurably less accurate than the single-vector cache of the same rank -- 0.75 against 0.57 of the noise at rank 100 -- so the
    reference-default rank 100 keeps the reference's recurrence; ``lanczos.block_size_for``), else 1.  Round 6: 32 from ``auto_wide_size`` rows on --
    there a product of up to 32 columns costs what one of 5 does (kernel generation bounds it: 45 ms at n = 500 000), so a rank-384 cache from 12
    products of 32 columns beats a rank-400 cache from 25 of 16: cold posterior 0.85 against 1.29 s, variance error 3.9e-4 against 5.0e-4 of the noise
    (``profiles/r06_s9_posterior_by_preconditioner_rank.json``)."""
    _global_value = "auto"
    auto_min_size = 16384
    auto_min_rank = 200
    auto_wide_size = 262144


class num_trace_samples(_value_context):
    _global_value = 10


class tridiagonal_jitter(_value_context):
    _global_value = 1e-6


class terminate_cg_by_size(_feature_flag):
    _default = False


class skip_logdet_forward(_feature_flag):
    _default = False


class rhs_refinement(_feature_flag):
    """(no counterpart in the reference.)  Mixed-precision iterative refinement of float32 solves: after float32 mBCG, the
    true residual rhs - K_hat a is formed with ONE float64 product (``csrc/kv_f64.hpp``, the same prepared points widened to float64), the
    correction K_hat d = r is solved in float32 as before and added -- ``rhs_refinement.steps`` times (default 1).  At kappa ~ 1e6 float32
    mBCG attains |a - a*| / |a*| ~ 3e-4 whatever the tolerance; one step takes that to the 1e-7 range, and with it the
    data-fit gradient -a^T dK a, the predictive mean and -- since round 5, any number of columns -- the n_test-column solve behind the EXACT
    predictive variance, whose last contraction then runs in float64 as well (``models.exact_predictive_covar``): the variance of f itself,
    1 - 0.9998.. at a well-determined point, instead of only the variance of y.  Cost: one float64 product with the solved columns + one more
    float32 solve.  OFF by default (the reference's float32 path has no such step); applies to the single-kernel float32 operator
    (d <= 16: the fused float64 kernel), unsharded rows."""
    _default = False
    steps = 1
    # Round 6: the n_test-column solve behind the EXACT predictive variance is not refined by a second solve any more (3.7 x the unrefined time);
    # the quadratic form is taken to second order in the solve error -- X^T (2 B - K_hat X), K_hat X in float64: ``bbmm.variational_inv_quad`` -- and
    # that error is the ENERGY norm of the solve error, which a residual tolerance bounds only through the condition number.  The float32 solve
    # therefore stops at ``variance_tolerance_factor`` x the tolerance in force: C2 (n = 100 000, 1000 test points, eval_cg_tolerance 1e-4) the
    # variance of f within rtol 2e-3 (+ the float32-input floor 2e-6) at 0.3 x (3.5 s against 2.6 s unrefined: 1.35 x; 0.9 x with the "auto"
    # preconditioner rank), not at 1.0 x (8 x the bound) -- profiles/r06_s6_posterior_at_size_c2_variational.json.
    variance_tolerance_factor = 0.3


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


class min_fixed_noise(_dtype_value_context):
    """``gpytorch/settings.py:272-296``: floor of FixedNoiseGaussianLikelihood noise values."""
    _global_float_value = 1e-4
    _global_double_value = 1e-6
    _global_half_value = 1e-3


class observation_nan_policy(_value_context):
    """``gpytorch/settings.py:416-446``: "ignore" | "mask" | "fill" -- how NaN observations are treated by the MLL."""
    _global_value = "ignore"

    def __init__(self, value):
        if value not in ("ignore", "mask", "fill"):
            raise ValueError(f"NaN handling policy {value} not supported!")
        super().__init__(value)


class fast_pred_samples(_feature_flag):
    """``gpytorch/settings.py:225-243``: draw posterior samples from the LOVE (Lanczos) root instead of a Cholesky factor."""
    _default = False


class num_contour_quadrature(_value_context):
    """``linear_operator.settings.num_contour_quadrature``: quadrature points of the CIQ rule (default 15)."""
    _global_value = 15


class ciq_samples(_feature_flag):
    """``gpytorch/settings.py`` ciq_samples: draw MVN samples as mu + K^{1/2} eps through contour-integral quadrature."""
    _default = False


class cholesky_jitter(_dtype_value_context):
    _global_float_value = 1e-6
    _global_double_value = 1e-8
    _global_half_value = 1e-4


class prior_mode(_feature_flag):
    """``gpytorch/settings.py:336-344``: evaluate an ExactGP in prior mode even if it has training data (``models/exact_gp.py:285``)."""
    _default = False


class minres_tolerance(_value_context):
    """Relative-residual tolerance of msMINRES under contour-integral quadrature (``linear_operator.settings.minres_tolerance``,
    re-exported by ``gpytorch/settings.py``; consumer ``gpytorch/__init__.py:252-278`` ``sqrt_inv_matmul``)."""
    _global_value = 1e-4


class cholesky_max_tries(_value_context):
    """Jitter levels ``psd_safe_cholesky`` tries before giving up (``linear_operator.settings.cholesky_max_tries``, re-exported at
    ``gpytorch/settings.py:11``): jitter x 10^i, i = 0 .. max_tries - 1."""
    _global_value = 3


class deterministic_probes(_feature_flag):
    """Re-use one fixed set of probe vectors (``linear_operator.settings.deterministic_probes``).
    ``deterministic_probes.probe_vectors`` may be pre-set to an (n, t) tensor to inject Z."""
    _default = False
    probe_vectors = None   # the most recently used / user-injected matrix
    _drawn = {}            # (n, t) -> matrix drawn by the library: alternating between models of different size keeps each one's probes
    max_kept = 4           # ... at most this many of them (insertion order; the oldest is dropped).  They deliberately SURVIVE the context --
    #                        a training loop that re-enters ``deterministic_probes(True)`` per closure call must see the same objective, as with
    #                        the reference's class attribute -- ``deterministic_probes.reset()`` releases them.

    @classmethod
    def reset(cls):
        """Forget the probe matrices the library drew (and the most recently used one)."""
        cls._drawn.clear()
        cls.probe_vectors = None


class cg_graph(_feature_flag):
    """(no counterpart in the reference.)  Small mBCG solves -- ``n^2 t <= cg_graph.max_work`` -- can record ONE iteration (fused
    K V, reduction, the vector updates, the preconditioner apply, the stopping rule) into a hipGraph after the first iteration
    and replay it.  Same kernels in the same order: the results are bitwise those of the eager loop
    (``test_cg_graph_replay_is_bitwise_the_eager_loop``).  OFF by default: measured on MI355X the eager loop is not launch-bound
    -- the host enqueues faster than the 7-15 dependent small kernels of an iteration execute (60 us per iteration at n = 2000,
    eleven columns) and the replayed graph runs them no faster (0.94-0.98x, ``profiles/r02_s28_cg_graph.json``)."""
    _default = False
    max_work = 1.0e10


class batched_small_members(_feature_flag):
    """(no counterpart in the reference, whose dense batch mode is batched by construction.)  A batch of independent exact GPs whose
    members are factorised (``n <= max_cholesky_size``) evaluates its marginal log likelihood with a launch count that does not depend
    on the batch size: ONE launch generates every member's dense covariance matrix, the factorisation / solves are torch's batched
    Cholesky, ONE launch reduces every member's kernel derivative (``gpytorch_amd/batched.py``, ``csrc/extra_batch.hip``).
    ``batched_small_members(False)`` keeps the launch plan over members (one Cholesky branch per member).
    Round 4: the stacked (exact, dense) evaluation also takes batches of MID-SIZE members -- ``max_cholesky_size < n <= max_size`` while the three
    float64 [b, n, n] work arrays stay within ``max_bytes`` -- which would otherwise run one BBMM evaluation per member: measured on MI355X
    (``profiles/r04_s18_batch_member_timing.json``, MLL + backward) 64 x 1000 points 137 -> 15.8 ms, 64 x 2000 237 -> 54 ms, break-even at
    n = 4000 (318 vs 302 ms).  The member loop is launch-bound there (about 2 ms per member whatever its size up to 2000 points); the dense
    factorisation is exact where mBCG + SLQ would be stochastic, so values only get MORE accurate.  ``max_size = 0`` restores the round-3 reach; a
    ``max_cholesky_size`` set BELOW its default (e.g. 0, to force mBCG) switches the extension off as well."""
    _default = True
    max_size = 3000
    max_bytes = 48e9


class split_contraction(_feature_flag):
    """(no counterpart in the reference.)  With five or more right-hand sides the fused ``K @ V`` contraction runs on the f16
    matrix pipe at float32 accuracy: both operands are split exactly into f16 hi + lo parts (21-22 significant bits, per-column
    power-of-two scaling, float32 accumulation) -- three ``v_mfma_f32_32x32x16_f16`` in place of eight
    ``v_mfma_f32_32x32x2_f32`` (``csrc/kv_gramh.hpp``; 2.5x at n = 500 000, 65 columns, same 2e-5 bound against the float64
    oracle).  ``split_contraction(False)`` keeps the contraction on the float32 MFMA instructions; the environment variable
    ``GPAMD_KV_SPLIT=0`` sets that as the process default."""
    import os as _os

    _default = _os.environ.get("GPAMD_KV_SPLIT", "1") not in ("0", "")


class far_pair_cutoff(_value_context):
    """Opt-in far-pair TILE CULLING of the fused float32 kernel products (default ``None`` = off: every pair is evaluated, the reference's
    arithmetic -- its KeOps seam reduces over all j, ``gpytorch/kernels/keops/rbf_kernel.py:44-55``).

    ``far_pair_cutoff(eps)``: with both clouds in Hilbert order, a 128-point tile of the contracted cloud is skipped for a block of output rows when
    the bounding spheres of the two are so far apart that EVERY covariance between them is <= ``eps`` (before the outputscale): it is neither loaded
    nor generated.  Every dropped entry of K is <= eps, so  |(K V)_ic - culled| <= eps * sum_j |V_jc|  per output entry -- with eps = 1e-7 below
    the float32 rounding of the entries that are kept.  Applies where the cloud is many lengthscales wide (block-centred Gram mode or the
    direct-difference kernels: short lengthscales, e.g. the first iterations of the reference's 3droad notebook at lengthscale 0.05,
    ``examples/02_Scalable_Exact_GPs/KeOps_GP_Regression.ipynb``); compact clouds and the heavy-tailed RQ family are not affected (nothing is far).
    The forward products (mBCG, posterior caches, matmul) are culled; the backward's bilinear derivative still visits every tile."""

    _global_value = None

    def __init__(self, value):
        if value is not None and not (0.0 < float(value) < 1.0):
            raise ValueError("far_pair_cutoff: eps must lie in (0, 1), or None to switch culling off")
        super().__init__(value)


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
        this group -- every rank ends up with the full caches;
      * ``mll_row_group``: the rows of the ``inv_quad_logdet`` solve are sharded over this group TOO (with ``probe_group``: the
        two-dimensional split of ``bbmm.inv_quad_logdet_forward`` -- probe groups x row blocks; the two groups must be orthogonal, i.e. a
        P x R grid of ranks).  For many GPUs and few probes per GPU (C4: 256 probes on 8 GPUs -> 4 x 2 keeps 64 + 1 columns per rank).

    ``sharding("auto")`` (or ``sharding(auto=True)``) asks for no layout at all -- what ``MultiDeviceKernel(base_kernel, device_ids)`` asks of its
    user (``kernels/multi_device_kernel.py:24-47``): the posterior's few-column solves are row-sharded over WORLD, and every MLL evaluation picks
    its P x R grid (probe shares x row blocks, P R = world size) from the number of points and probes with the cost model of
    ``distributed.choose_grid`` (measured column ladder of the fused K*V + one all-gather of the search directions per iteration over a row
    group) and builds the subgroups on first use (``distributed.grid_groups``: collective, cached).  The metric workload (n = 500 000, 64
    probes) on 8 GPUs: 1 x 8 -- 65 columns on an eighth of the rows each; C4 (n = 10^6, 256 probes): 2 x 4.

    Replaces ``gpytorch.kernels.MultiDeviceKernel`` (``multi_device_kernel.py:49-92``).  Not thread-safe, like every
    other setting here (``gpytorch/settings.py:84-144`` are process-global class attributes)."""

    _probe_group = None
    _row_group = None
    _mll_row_group = None
    _auto = False
    _generators: dict = {}

    def __init__(self, probe_group=None, row_group=None, mll_row_group=None, auto=False):
        if isinstance(probe_group, str):
            if probe_group != "auto":
                raise ValueError(f"settings.sharding: unknown policy {probe_group!r} (process groups, or \"auto\")")
            probe_group, auto = None, True
        if auto and (probe_group is not None or row_group is not None or mll_row_group is not None):
            raise ValueError("settings.sharding: 'auto' chooses the groups itself; pass either groups or auto")
        self._new = (probe_group, row_group, mll_row_group, bool(auto))

    @classmethod
    def is_auto(cls) -> bool:
        import torch.distributed as dist

        return bool(cls._auto) and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    @classmethod
    def mll_groups(cls, n: int, t_total: int, allow_rows: bool = True):
        """(probe group, row group) of one MLL evaluation with ``t_total`` probes on ``n`` points: the explicit groups of the scope, or -- under
        ``"auto"`` -- the grid ``distributed.choose_grid`` picks (same arguments on every rank -> same grid, same collective group creation)."""
        if not cls.is_auto():
            return cls._probe_group, (cls._mll_row_group if allow_rows else None)
        import torch.distributed as dist

        from .distributed import choose_grid, grid_groups

        P, R = choose_grid(dist.get_world_size(), int(n), int(t_total), "split" if split_contraction.on() else "f32", allow_rows=allow_rows)
        return grid_groups(P, R)

    @classmethod
    def mll_row_group(cls):
        return cls._mll_row_group

    @classmethod
    def probe_group(cls):
        return cls._probe_group

    @classmethod
    def row_group(cls):
        if cls._row_group is None and cls.is_auto():
            import torch.distributed as dist

            return dist.group.WORLD
        return cls._row_group

    @classmethod
    def rank_generator(cls, group, device):
        """The probe generator of this rank: created ONCE per (group, device) from the process seed and the rank, then
        left to advance -- consecutive MLL evaluations draw fresh probes (as the single-GPU path does from the global
        RNG) while ranks stay decorrelated.  ``torch.manual_seed`` followed by ``sharding.reset_generators()`` re-seeds."""
        key = (id(group), str(device))
        gen = cls._generators.get(key)
        if gen is None:
            rank = torch.distributed.get_rank(group)
            gen = torch.Generator(device=device).manual_seed(torch.initial_seed() % (2**31) + 7919 * (rank + 1))
            cls._generators[key] = gen
        return gen

    @classmethod
    def reset_generators(cls):
        cls._generators.clear()

    def __enter__(self):
        self._old = (sharding._probe_group, sharding._row_group, sharding._mll_row_group, sharding._auto)
        sharding._probe_group, sharding._row_group, sharding._mll_row_group, sharding._auto = self._new
        return self

    def __exit__(self, *args):
        sharding._probe_group, sharding._row_group, sharding._mll_row_group, sharding._auto = self._old
        return False
