"""Modified batched conjugate gradients (mBCG) -- host driver of the device-resident solver.

Mirrors ``linear_operator.utils.linear_cg`` (third-party; restated in ``oracle/linear_cg.py`` and
SURVEY.md A.2): same normalisation of the right-hand sides, same per-column alpha/beta masks, same
stopping rule (>= 10 iterations, mean relative residual < tolerance, and >= the Lanczos-quadrature
floor when tridiagonals are requested), same tridiagonal-matrix formulas.  Reference call sites:
``gpytorch/distributions/multivariate_normal.py:249`` and
``gpytorch/models/exact_prediction_strategies.py:286,444``.

Differences, all by design (DESIGN.md "mBCG"):
  * state is probe-major ``[t, ld]`` and lives on the GPU; one iteration is 4-5 kernel launches
    (fused K*V partials, reduce+dot, x/r update, d update, stop test) instead of ~15 torch ops;
  * the host does not synchronise per iteration: it enqueues ahead and polls one ``done`` word;
    launches issued after convergence are device-side no-ops, so results are identical to a
    per-iteration check;
  * with probe columns sharded over ranks, the only communication is an RCCL all-reduce of two
    floats (sum of residual norms, column count) per iteration for the global stopping rule.
"""
from __future__ import annotations

import ctypes as C
import warnings
from dataclasses import dataclass

import torch

from . import backend as B
from .distributed import allreduce_max_int, allreduce_sum_
from . import settings
from ._lib import check, lib


# When set to a list, every fused K*V launch of the CG loop is bracketed by HIP events on the launch
# stream and (start, end, n, t) is appended -- bench.py uses it to time the dominant kernel live.
KV_EVENT_LOG: list | None = None
LAST_INFO = None  # CGInfo of the most recent solve in this process (diagnostics / tests)


class NumericalWarning(RuntimeWarning):
    """Same role as ``gpytorch.utils.warnings.NumericalWarning``."""


@dataclass
class CGInfo:
    iterations: int
    tolerance_reached: bool
    residual_norms: torch.Tensor  # [t] relative residual norms (device)
    t_mats: torch.Tensor | None   # [n_tridiag, m, m] (CPU float64) or None


def build_tridiag(alpha: torch.Tensor, beta: torch.Tensor, iters: int, broke: bool, n_tri_iter: int) -> torch.Tensor:
    """T matrices from the CG coefficient history (linear_cg's tridiagonal update).

    alpha, beta: [hist, n_tridiag] (CPU).  Rows exist for k < min(n_tri_iter, iters - broke)."""
    nt = alpha.shape[1]
    rows = min(n_tri_iter, iters - (1 if broke else 0), alpha.shape[0])
    T = torch.zeros(nt, max(rows, 1), max(rows, 1), dtype=torch.float64)
    update = True
    last = 0
    ainv_prev = b_prev = None
    for k in range(rows):
        if not update:
            break
        a = alpha[k].to(torch.float64)
        b = beta[k].to(torch.float64)
        ainv = 1.0 / torch.where(a == 0, torch.ones_like(a), a)
        if k == 0:
            T[:, 0, 0] = ainv
        else:
            T[:, k, k] = ainv + b_prev * ainv_prev
            off = b_prev.sqrt() * ainv_prev
            T[:, k, k - 1] = off
            T[:, k - 1, k] = off
            if float(off.max()) < 1e-6:
                update = False
        last = k
        ainv_prev, b_prev = ainv, b
    return T[:, : last + 1, : last + 1].contiguous()


class Preconditioner:
    """P = L L^T + s2 I applied through the thin QR of [L; sqrt(s2) I].

    ``q1t``: [k, ld] (rows = columns of Q1), ``sigma2``: 1-element device tensor."""

    def __init__(self, q1t: torch.Tensor, sigma2: torch.Tensor, logdet: torch.Tensor, lt: torch.Tensor, reduce=None, dinv_sqrt=None):
        self.q1t, self.sigma2, self.logdet, self.lt = q1t, sigma2, logdet, lt
        self.reduce = reduce  # row-sharded: in-place sum over ranks of the k x t coefficients R Q1
        # NON-constant diagonal D (fixed heteroskedastic noise, per-task noise of a multitask likelihood; the reference's
        # ``_init_cache_for_non_constant_diag``): P = L L^T + D = D^1/2 (Lt Lt^T + I) D^1/2 with Lt = D^-1/2 L, so Q1 is the thin-QR
        # factor of [Lt; I], ``sigma2`` is 1 and the apply is D^-1/2 (I - Q1 Q1^T) D^-1/2.  ``dinv_sqrt``: [ld] (zeros beyond n), or None.
        self.dinv_sqrt = dinv_sqrt

    def noise_sqrt(self, n: int, dtype):
        """sqrt of the diagonal part of P as something that broadcasts against [t, n] (probe sampling z = L e1 + D^1/2 e2)."""
        if self.dinv_sqrt is None:
            return self.sigma2.to(dtype).sqrt()
        return (1.0 / self.dinv_sqrt[:n]).to(dtype).unsqueeze(0)

    def row_sharded(self, row_shard) -> "Preconditioner":
        """The same preconditioner acting on vectors of which every rank holds a block of rows: Q1 is sliced to the local
        rows and the k x t inner products R Q1 are all-reduced per apply (k <= 512, t small: a few KB over xGMI)."""
        di = None if self.dinv_sqrt is None else row_shard.local(self.dinv_sqrt.unsqueeze(0))[0]
        return Preconditioner(row_shard.local(self.q1t), self.sigma2, self.logdet, row_shard.local(self.lt), reduce=row_shard.allreduce, dinv_sqrt=di)

    def apply_(self, rt: torch.Tensor, out: torch.Tensor):
        """out = (R - (R Q1) Q1^T) / s2 in probe-major form (rows are vectors), evaluated in FLOAT64 whatever the dtype of the
        solve.  I - Q1 Q1^T has eigenvalues s2 / (s2 + lambda): for a smooth kernel (RBF, d = 3, n = 5e5: lambda / s2 ~ 1e6) the
        leading components of R must cancel to 6 digits, which float32 cannot deliver -- the float32 apply made preconditioned CG
        stall at a relative residual of 4 after 2000 iterations (profiles/r02_s5_posterior_profile_precond_fp32_stalls.json).
        Cost: two kernels per CG iteration (the k x t coefficients, then the fused subtraction / division), < 1 ms against a
        >= 19 ms K*V at n = 5e5 and ~25 us at n = 2000 (the four torch float64 operations they replaced took 280 us there)."""
        if self.dinv_sqrt is not None:
            di = self.dinv_sqrt[: rt.shape[1]].to(rt.dtype).unsqueeze(0)
            self._apply_core(rt * di, out)
            out[:, : di.shape[1]].mul_(di)
            return out
        return self._apply_core(rt, out)

    def _apply_core(self, rt: torch.Tensor, out: torch.Tensor):
        q1t = self.q1t if self.q1t.dtype == torch.float64 else self.q1t.to(torch.float64)
        k = q1t.shape[0]
        fast = rt.dtype == torch.float32 and rt.is_cuda and k <= 512 and rt.stride(1) == 1 and q1t.stride(1) == 1
        if fast:
            # W = R Q1^T: own mixed-precision reduction kernel (rocBLAS' float64 GEMM takes 80 ms for some tall-skinny shapes)
            t, n = rt.shape[0], min(rt.shape[1], q1t.shape[1])
            L = lib()
            key = (t, n, k, rt.device)
            if getattr(self, "_apply_ws_key", None) != key:
                nws = int(L.gpamd_precond_coef_workspace_doubles(n, t, k))
                self._apply_ws = (torch.empty(nws, device=rt.device, dtype=torch.float64), torch.empty(t, k, device=rt.device, dtype=torch.float64))
                self._apply_ws_key = key
            ws, w = self._apply_ws
            check(L.gpamd_precond_coef_f32f64(B._ptr(rt), rt.stride(0), t, B._ptr(q1t), q1t.stride(0), k, n, B._ptr(w), B._ptr(ws), ws.numel(),
                                              B._stream(rt.device)), "precond_coef")
            if self.reduce is not None:
                self.reduce(w)
            s2 = self.sigma2 if self.sigma2.dtype == torch.float32 else self.sigma2.to(torch.float32)
            if out.dtype == torch.float32 and out.stride(1) == 1 and out.shape[1] >= n and out.data_ptr() != rt.data_ptr() and s2.numel() == 1:
                # second half fused: (R - W Q1) / s2 in float64 arithmetic, one kernel, no float64 temporaries
                check(L.gpamd_precond_apply_f32f64(B._ptr(rt), rt.stride(0), t, B._ptr(q1t), q1t.stride(0), k, n, B._ptr(w), B._ptr(s2),
                                                   B._ptr(out), out.stride(0), B._stream(rt.device)), "precond_apply")
                if out.shape[1] > n:
                    out[:, n:].zero_()
                return out
            r64 = rt.to(torch.float64)
        else:
            r64 = rt.to(torch.float64)
            w = r64 @ q1t.t()
            if self.reduce is not None:
                self.reduce(w)
        z = torch.addmm(r64, w, q1t, alpha=-1.0)
        z.div_(self.sigma2.to(torch.float64))
        out.copy_(z)
        return out


def linear_cg(
    x: B.PreparedPoints,
    scale: torch.Tensor | None,
    dscale: torch.Tensor | None,
    rhs_t: torch.Tensor,
    n_tridiag: int = 0,
    tolerance: float | None = None,
    eps: float = 1e-10,
    stop_updating_after: float = 1e-10,
    max_iter: int | None = None,
    max_tridiag_iter: int | None = None,
    preconditioner: Preconditioner | None = None,
    group=None,
    kv_partials=None,
    dvec: torch.Tensor | None = None,
    nvec: int | None = None,
    row_shard=None,
):
    """Solve (scale*K(x,x) + dscale*I + diag(dvec)) X = rhs for all rows of ``rhs_t`` ([t, ld], probe-major).

    ``kv_partials(Dt) -> (P, S, ldp)``: optional override of the noise-free operator product (used by the
    multitask Kronecker operator, whose vectors have length n*T): returns partial slabs ``P[S][t][ldp]`` whose
    sum is ``K_op @ D``; ``x`` may then be ``None`` and the vector length is taken from ``nvec``.
    ``row_shard`` (:class:`gpytorch_amd.distributed.RowShard`): this rank owns a block of ROWS of the system; ``rhs_t``,
    ``dvec`` and the returned solves are its local slices, products gather the search directions over ranks, and the
    solver's inner products are all-reduced (float32; a full-length ``preconditioner`` is sliced to the local rows and its
    k x t coefficients are all-reduced per apply).
    Returns (solves_t [t, ld], CGInfo)."""
    B._require_gpu(rhs_t, "rhs")
    L = lib()
    if row_shard is not None:
        if rhs_t.dtype != torch.float32 or kv_partials is not None:
            raise NotImplementedError("row-sharded solves: float32, fused kernel operator")
        if preconditioner is not None and preconditioner.reduce is None:
            preconditioner = preconditioner.row_sharded(row_shard)
        x, nvec = None, row_shard.n_loc

        def kv_partials(dt_, _rs=row_shard):
            out = _rs.kv_local(dt_)  # unscaled K[rows, :] @ D; scale / noise are applied by reduce_q
            return out, 1, out.stride(0)
    n = x.n if nvec is None else nvec
    if x is not None and rhs_t.dtype != x.dtype:
        rhs_t = rhs_t.to(x.dtype)  # the solve runs in the dtype of the prepared points
    elif rhs_t.dtype not in (torch.float32, torch.float64):
        rhs_t = rhs_t.to(torch.float32)
    t, ld = rhs_t.shape
    dev = rhs_t.device
    if tolerance is None:
        tolerance = settings.cg_tolerance.value()
    if max_iter is None:
        max_iter = settings.max_cg_iterations.value()
    if max_tridiag_iter is None:
        max_tridiag_iter = settings.max_lanczos_quadrature_iterations.value()
    n_tri_iter = min(max_tridiag_iter, n)
    hist = min(n_tri_iter, max_iter) if n_tridiag else 0

    st = B._stream(dev)
    # the vector kernels are templated on the scalar type
    f64 = rhs_t.dtype == torch.float64
    wd = torch.float64 if f64 else torch.float32
    F = {k: getattr(L, (f"gpamd_cg64_{k}" if f64 else f"gpamd_cg_{k}_f32"))
         for k in ("init", "begin", "reduce_q", "update_xr", "update_d", "stop", "finish")}
    Xt = torch.zeros(t, ld, device=dev, dtype=wd)
    Rt = torch.zeros_like(Xt)
    Dt = torch.zeros_like(Xt)
    Qt = torch.zeros_like(Xt)
    Zt = torch.zeros_like(Xt) if preconditioner is not None else Rt
    fs = torch.zeros(int(L.gpamd_cg_fscratch_elems(t, hist)), device=dev, dtype=wd)
    isc = torch.zeros(int(L.gpamd_cg_iscratch_elems(t)), device=dev, dtype=torch.int32)
    offs = (C.c_int64 * 5)()
    check(L.gpamd_cg_layout(t, hist, offs), "cg_layout")
    create = L.gpamd_cg64_create if f64 else L.gpamd_cg_create_f32
    h = create(n, t, ld, B._ptr(Xt), B._ptr(Rt), B._ptr(Dt), B._ptr(Qt), B._ptr(Zt), B._ptr(fs), B._ptr(isc), hist,
               float(eps), float(stop_updating_after))
    if not h:
        check(-1, "cg_create")
    done_t = isc[2 * t : 2 * t + 2]
    done_ptr = C.c_void_p(isc.data_ptr() + 2 * t * 4)
    stats = fs[offs[4] : offs[4] + 2]
    scale = None if scale is None else scale.to(wd)
    dscale = None if dscale is None else dscale.to(wd)
    dvec = None if dvec is None else dvec.to(wd)
    if kv_partials is None and not x.fused:
        # generic path: fused float64 kernel (d <= 8), else dense row blocks of K (HIP) x library GEMM as one "partial" slab
        if B.fused_f64(x, x):
            def kv_partials(dt_, _x=x):
                return B.kv_partials_f64(_x, _x, dt_, done_ptr)
        else:
            def kv_partials(dt_, _x=x):
                out = B.kv_chunked(_x, _x, dt_)
                return out, 1, out.stride(0)
    if row_shard is not None:
        po = (C.c_int64 * 3)()
        pstride, pnb = C.c_int(), C.c_int()
        check(L.gpamd_cg_partials_layout(n, t, hist, po, C.byref(pstride), C.byref(pnb)), "cg_partials_layout")

        def ar(which):
            row_shard.allreduce_partials(fs, int(po[which]), t, pstride.value)
    try:
        if row_shard is not None:
            check(L.gpamd_cg_init_norms_f32(h, B._ptr(rhs_t), rhs_t.stride(0), st), "cg_init_norms")
            ar(0)
            check(L.gpamd_cg_init_apply_f32(h, B._ptr(rhs_t), rhs_t.stride(0), 0 if preconditioner is not None else 1, st), "cg_init_apply")
            ar(2)
            if preconditioner is None:
                ar(1)
            else:
                preconditioner.apply_(Rt, Zt)
                Dt.copy_(Zt)
                check(L.gpamd_cg_dot_rz_f32(h, st), "cg_dot_rz")
                ar(1)
            check(L.gpamd_cg_begin_apply_f32(h, st), "cg_begin_apply")
        else:
            check(F["init"](h, B._ptr(rhs_t), rhs_t.stride(0), 1 if preconditioner is not None else 0, st), "cg_init")
            if preconditioner is not None:
                preconditioner.apply_(Rt, Zt)
                Dt.copy_(Zt)
                check(F["begin"](h, st), "cg_begin")

        if kv_partials is None:
            flags = B.kv_flags(x, x, t)
            S, jc, wsn = B.kv_plan(x.kind, n, n, x.d, t, flags, ld)
            P = B.workspace(dev, wsn)
            if B.rows_sorted(x, x, flags):
                Psum = torch.zeros(t, ld, device=dev, dtype=torch.float32)
                Pq1 = torch.zeros(t, ld, device=dev, dtype=torch.float32)
        ldp = ld
        min_iter = min(10, max_iter - 1)
        tri_floor = min(n_tri_iter, max_iter - 1) if n_tridiag else 0
        first_poll = max(min_iter, tri_floor)
        # (row-sharded: the schedule must not depend on the local row count -- a rank that stops polling-late would issue
        # collectives its peers never join)
        # sharded solves: every rank must leave the loop at the SAME iteration (a rank that polls later would issue collectives
        # its peers never join), so the polling schedule is built from rank-independent quantities only -- the padded local row
        # count and the LARGEST column count of any rank.  The stopping statistics themselves are all-reduced on the device
        # stream (RCCL), so the done flag flips at the same iteration everywhere and nobody needs to poll every iteration.
        n_poll = row_shard.n_pad if row_shard is not None else n
        t_poll = allreduce_max_int(t, group) if group is not None else t
        poll_every = 1 if float(n_poll) * n_poll * t_poll > 2e11 else 8
        flag = 0
        iters = 0

        def iteration(k, st):
            """One mBCG iteration on stream ``st``; k = -1: the kernels take the iteration index from the device (graph replay)."""
            nonlocal P, S, ldp
            ev = None
            if KV_EVENT_LOG is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record(torch.cuda.current_stream(dev))
            if kv_partials is None:
                Pq, Sq = P, S
                unsort = B.kv_partials_sorted(x, x, Dt, t, flags, P, ld, S, jc, done_ptr, st)
                if unsort is not None:
                    # block-centred Gram expansion (wide clouds): the slabs hold the rows in Hilbert order -> sum them, take the rows back to
                    # the original order (t x n floats per iteration against n^2 t pair evaluations), hand ONE slab to the solver
                    check(L.gpamd_kv_reduce_f32(B._ptr(P), S, ld, t, n, None, None, None, None, 0, B._ptr(Psum), ld, done_ptr, st), "kv_reduce")
                    torch.index_select(Psum, 1, unsort, out=Pq1)
                    Pq, Sq = Pq1, 1
            else:
                P, S, ldp = kv_partials(Dt)
                Pq, Sq = P, S
            if ev is not None:
                ev[1].record(torch.cuda.current_stream(dev))
                KV_EVENT_LOG.append((ev[0], ev[1], n, t, k))
            check(F["reduce_q"](h, B._ptr(Pq), Sq, ldp, B._ptr(scale), B._ptr(dscale), B._ptr(dvec), st), "cg_reduce_q")
            if row_shard is not None:
                ar(0)
            check(F["update_xr"](h, k, st), "cg_update_xr")
            if row_shard is not None:
                ar(2)
                if preconditioner is None:
                    ar(1)
                else:
                    preconditioner.apply_(Rt, Zt)
                    check(L.gpamd_cg_dot_rz_f32(h, st), "cg_dot_rz")
                    ar(1)
                check(L.gpamd_cg_update_d_apply_f32(h, k, st), "cg_update_d_apply")
            else:
                if preconditioner is not None:
                    preconditioner.apply_(Rt, Zt)
                check(F["update_d"](h, k, st), "cg_update_d")
            if group is not None:
                allreduce_sum_(stats, group)
            check(F["stop"](h, k, min_iter, tri_floor, float(tolerance), st), "cg_stop")

        # settings.cg_graph (off by default: measured, no gain -- the loop is bound by the execution of its small dependent kernels,
        # not by their launches): iteration 0 runs eagerly, then ONE iteration is recorded into a hipGraph (torch's stream capture
        # sees the ctypes launches like any other work on the capturing stream) and replayed; the kernels read the iteration index
        # from the device, converged solves keep turning into no-ops through the done flag as before.
        graph = None
        use_graph = (settings.cg_graph.on() and kv_partials is None and x.fused and group is None and row_shard is None
                     and KV_EVENT_LOG is None and max_iter > 2 and float(n) * n * t <= settings.cg_graph.max_work)
        for k in range(max_iter):
            if graph is not None:
                graph.replay()
            else:
                iteration(k, st)
                if use_graph and k == 0:
                    try:
                        graph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(graph):
                            iteration(-1, B._stream(dev))
                    except Exception as exc:   # capture refused (an op in a user-supplied preconditioner, an old runtime): stay eager
                        warnings.warn(f"mBCG: graph capture failed ({exc}); continuing with eager launches", RuntimeWarning)
                        graph, use_graph = None, False
                        torch.cuda.synchronize(dev)
            iters = k + 1
            if k >= first_poll and ((k - first_poll) % poll_every == 0 or k == max_iter - 1):
                flag, iters_dev = (int(v) for v in done_t.tolist())
                if flag:
                    iters = iters_dev
                    break
        if not flag:
            flag, iters_dev = (int(v) for v in done_t.tolist())
            iters = iters_dev if flag else max_iter
        if flag == 2:
            raise RuntimeError("NaNs encountered when trying to perform matrix-vector multiplication")
        check(F["finish"](h, st), "cg_finish")
        rnorm = fs[offs[1] : offs[1] + t].clone()
        tolerance_reached = flag == 1
        if not tolerance_reached and max_iter > 0:
            warnings.warn(
                f"CG terminated in {iters} iterations with average residual norm {float(rnorm.mean())} "
                f"which is larger than the tolerance of {tolerance} specified by "
                "gpytorch_amd.settings.cg_tolerance. If performance is affected, consider raising the "
                "maximum number of CG iterations by running code in a "
                "gpytorch_amd.settings.max_cg_iterations(value) context.",
                NumericalWarning,
            )
        t_mats = None
        if n_tridiag:
            a_h = fs[offs[2] : offs[2] + hist * t].view(hist, t)[:, :n_tridiag].cpu()
            b_h = fs[offs[3] : offs[3] + hist * t].view(hist, t)[:, :n_tridiag].cpu()
            t_mats = build_tridiag(a_h, b_h, iters, tolerance_reached, n_tri_iter)
    finally:
        (L.gpamd_cg64_destroy if f64 else L.gpamd_cg_destroy)(h)
    global LAST_INFO
    LAST_INFO = CGInfo(iters, tolerance_reached, rnorm, t_mats)
    return Xt, LAST_INFO
