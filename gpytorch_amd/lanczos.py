"""Lanczos tridiagonalisation with full re-orthogonalisation and the root / root-inverse
decompositions built on it (LOVE predictive-variance cache).

Mirrors ``linear_operator.utils.lanczos.lanczos_tridiag`` / ``lanczos_tridiag_to_diag`` and
``LinearOperator.root_inv_decomposition`` (third-party; restated in ``oracle/lanczos.py``,
SURVEY.md A.7).  Reference call sites: ``gpytorch/models/exact_prediction_strategies.py:202,
234-238,271``.  Each step is one fused K_hat*q (t = 1: generation-bound) plus O(n k) re-orthogonalisation
on probe-major rows.  float32 vectors run on the HIP kernels of ``csrc/lanczos_kernels.hpp`` (``gpamd_lanczos_*``:
wave-shuffle reductions, alpha / beta / projection coefficients resident on the device, ONE host poll per step for the
re-orthogonalisation / breakdown flags); float64 (generic path) keeps the torch formulation of the same algorithm.
"""
from __future__ import annotations

import torch

from . import backend as B
from . import settings


def lanczos_tridiag(x: B.PreparedPoints, scale, dscale, max_iter: int, init_vec_t: torch.Tensor | None = None,
                    tol: float = 1e-5, generator=None, dvec=None, matvec=None, nvec=None, device=None, reduce=None, n_global=None):
    """Dispatch: float32 -> device-resident step kernels (:func:`_lanczos_native`), float64 -> :func:`_lanczos_torch`."""
    wd = init_vec_t.dtype if init_vec_t is not None else (x.dtype if x is not None else torch.float32)
    fn = _lanczos_native if (wd == torch.float32 and not FORCE_TORCH and max_iter <= 512) else _lanczos_torch  # (kernels: k <= 512)
    return fn(x, scale, dscale, max_iter, init_vec_t, tol, generator, dvec, matvec, nvec, device, reduce, n_global)


FORCE_TORCH = False  # tests: run float32 problems through the torch formulation too


def lanczos_steps(n, dev, max_iter, init_vec_t=None, tol=1e-5, generator=None, reduce=None, n_global=None):
    """The algorithm of :func:`_lanczos_torch` as a COROUTINE, vector work on the ``gpamd_lanczos_*`` kernels: yields the
    probe-major row q_k ([1, ld]) whose product w = A q_k it needs next and receives w through ``send``; finally returns
    (Qt [m, ld], T [m, m]) as the StopIteration value.  The caller owns the operator product -- which lets the predictive
    posterior fuse it with the mean-cache CG product into one two-column launch (:func:`gpytorch_amd.models.fused_caches`)."""
    import ctypes as C

    from ._lib import check, lib

    ld = B.round_up(n, 4)
    num_iter = min(max_iter, n if n_global is None else n_global)
    f32 = torch.float32
    if init_vec_t is None:
        init_vec_t = torch.zeros(1, ld, device=dev, dtype=f32)
        init_vec_t[:, :n] = torch.randn(1, n, device=dev, generator=generator, dtype=f32)
    init_vec_t = init_vec_t.to(f32).contiguous()
    L, st = lib(), B._stream(dev)
    nb, stride = int(L.gpamd_lanczos_num_partials(n)), int(L.gpamd_lanczos_partial_stride())
    Q = torch.zeros(num_iter, ld, device=dev, dtype=f32)
    alpha = torch.zeros(num_iter, device=dev, dtype=f32)
    beta = torch.zeros(num_iter, device=dev, dtype=f32)
    part = torch.empty(max(num_iter, 1) * stride, device=dev, dtype=f32)
    coef = torch.empty(max(num_iter, 1), device=dev, dtype=f32)
    rr = torch.empty(1, device=dev, dtype=f32)
    flags = torch.zeros(2, device=dev, dtype=torch.int32)  # [0]: some |<Q, r>| > tol   [1]: breakdown (beta < 1e-6)
    r = torch.zeros(1, ld, device=dev, dtype=f32)
    p = B._ptr
    flag_ptr = C.c_void_p(flags.data_ptr())
    stop_ptr = C.c_void_p(flags.data_ptr() + 4)

    def fptr(t_, off=0):
        return C.c_void_p(t_.data_ptr() + 4 * off)

    def project(basis, k, out, out_off=0, tol_=-1.0):
        """out[out_off : out_off + k] = <basis[m], r> (global when sharded); tol_ >= 0 raises the re-orthogonalisation flag."""
        check(L.gpamd_lanczos_project_f32(p(basis), basis.stride(0), k, p(r), n, p(part), st), "lanczos_project")
        dev_flag = tol_ >= 0 and reduce is None
        check(L.gpamd_lanczos_coef_f32(p(part), k, nb, tol_ if dev_flag else -1.0, fptr(out, out_off), flag_ptr if dev_flag else None, st), "lanczos_coef")
        if reduce is not None:
            reduce(out[out_off : out_off + k])
            if tol_ >= 0:
                flags[0] = (out[out_off : out_off + k].abs() > tol_).any().to(torch.int32)

    def subtract(basis, k, cf, cf_off=0):
        """r -= sum_m cf[m] basis[m];  rr = |r|^2 (global when sharded)."""
        check(L.gpamd_lanczos_subtract_f32(p(basis), basis.stride(0), k, fptr(cf, cf_off), p(r), n, p(part), st), "lanczos_subtract")
        check(L.gpamd_lanczos_coef_f32(p(part), 1, nb, -1.0, p(rr), None, st), "lanczos_coef")
        if reduce is not None:
            reduce(rr)

    def normalize(src, dst_row, norm_out=None, norm_off=0, watch=False):
        check(L.gpamd_lanczos_normalize_f32(p(src), n, p(rr), p(dst_row), None if norm_out is None else fptr(norm_out, norm_off),
                                            1e-6, stop_ptr if watch else None, st), "lanczos_normalize")

    # q0 = init / |init|
    r.copy_(init_vec_t[:1])
    zero = torch.zeros(1, device=dev, dtype=f32)
    subtract(Q[0:1], 1, zero)                      # r unchanged, rr = |init|^2
    normalize(r, Q[0])
    w = yield Q[0:1]
    check(L.gpamd_lanczos_residual_f32(p(w), None, None, p(r), n, st), "lanczos_residual")
    project(Q[0:1], 1, alpha, 0)                   # alpha_0 = <q0, K q0>
    subtract(Q[0:1], 1, alpha, 0)                  # r -= alpha_0 q0, rr = |r|^2
    m = 1
    if num_iter > 1:
        normalize(r, Q[1], beta, 0)                # beta_0 = |r|, q1 = r / beta_0
        m = 2
        for k in range(1, num_iter):
            w = yield Q[k : k + 1]
            check(L.gpamd_lanczos_residual_f32(p(w), p(Q[k - 1]), fptr(beta, k - 1), p(r), n, st), "lanczos_residual")
            project(Q[k : k + 1], 1, alpha, k)     # alpha_k = <q_k, r>
            m = k + 1
            if k + 1 >= num_iter:
                break
            subtract(Q[k : k + 1], 1, alpha, k)
            project(Q, k + 1, coef)                # full re-orthogonalisation against q_0 .. q_k
            subtract(Q, k + 1, coef)
            normalize(r, r, beta, k, watch=True)   # beta_k = |r|, r /= beta_k; breakdown flag
            ok = False
            for _ in range(10):
                flags[0:1].zero_()
                project(Q, k + 1, coef, 0, tol_=tol)
                need, stop = (int(v) for v in flags.tolist())   # the one host poll of this pass
                if not need:
                    ok = True
                    break
                subtract(Q, k + 1, coef)
                normalize(r, r)
            Q[k + 1].copy_(r[0])
            if stop or not ok:
                break
            m = k + 2
    a_h, b_h = alpha[:m], beta[: max(m - 1, 0)]
    T = torch.diag(a_h)
    if m > 1:
        T = T + torch.diag(b_h, 1) + torch.diag(b_h, -1)
    return Q[:m], T


def _lanczos_native(x, scale, dscale, max_iter, init_vec_t, tol, generator, dvec, matvec, nvec, device, reduce, n_global):
    """Drives :func:`lanczos_steps` with this operator's own product.  Same return convention as :func:`_lanczos_torch`."""
    n = x.n if nvec is None else nvec
    dev = x.xp.device if device is None else device

    def mv(q_row):
        if matvec is not None:
            return matvec(q_row)
        return B.kv(x, x, q_row, scale=scale, dscale=dscale, vd=q_row if dscale is not None else None, dvec=dvec)

    steps = lanczos_steps(n, dev, max_iter, init_vec_t, tol, generator, reduce, n_global)
    try:
        q = next(steps)
        while True:
            q = steps.send(mv(q))
    except StopIteration as done:
        return done.value


def _lanczos_torch(x, scale, dscale, max_iter, init_vec_t, tol, generator, dvec, matvec, nvec, device, reduce, n_global):
    """Returns (Qt [m, ld] with orthonormal rows, T [m, m] on device, in the dtype of the prepared points).

    ``matvec(q_row [1, ld]) -> [1, ld]``: optional operator override (multitask Kronecker); then ``x`` may be
    None and ``nvec`` / ``device`` give the vector length and device.
    ``reduce(tensor)``: optional in-place sum over ranks (row-sharded vectors, :class:`distributed.RowShard`): every
    inner product / norm / projection below is then a global one and T is identical on all ranks."""
    n = x.n if nvec is None else nvec
    dev = x.xp.device if device is None else device
    ld = B.round_up(n, 4)
    num_iter = min(max_iter, n if n_global is None else n_global)  # row-sharded: the same step count on every rank
    wd = init_vec_t.dtype if init_vec_t is not None else (x.dtype if x is not None else torch.float32)
    if init_vec_t is None:
        init_vec_t = torch.zeros(1, ld, device=dev, dtype=wd)
        init_vec_t[:, :n] = torch.randn(1, n, device=dev, generator=generator, dtype=wd)
    Q = torch.zeros(num_iter, ld, device=dev, dtype=wd)
    T = torch.zeros(num_iter, num_iter, device=dev, dtype=wd)

    def rsum(v):  # (global) sum of all entries, 0-dim
        s_ = v.sum()
        return reduce(s_) if reduce is not None else s_

    def rnorm(v):
        return rsum(v * v).sqrt()

    def proj(v, basis_):  # (global) coefficients of v in the rows of basis_
        c_ = v @ basis_.t()
        return reduce(c_) if reduce is not None else c_

    def mv(q_row):  # K_hat q, q_row: [1, ld]
        if matvec is not None:
            return matvec(q_row)
        return B.kv(x, x, q_row, scale=scale, dscale=dscale, vd=q_row if dscale is not None else None, dvec=dvec)

    q0 = init_vec_t / rnorm(init_vec_t)
    Q[0] = q0[0]
    r = mv(q0)
    a0 = rsum(q0 * r)
    r = r - a0 * q0
    b0 = rnorm(r)
    T[0, 0] = a0
    m = 1
    if num_iter > 1:
        T[0, 1] = b0
        T[1, 0] = b0
        Q[1] = (r / b0)[0]
        m = 2
        for k in range(1, num_iter):
            q_prev, q = Q[k - 1 : k], Q[k : k + 1]
            b_prev = T[k, k - 1]
            r = mv(q) - b_prev * q_prev
            a = rsum(q * r)
            T[k, k] = a
            m = k + 1
            if k + 1 < num_iter:
                r = r - a * q
                basis = Q[: k + 1]
                r = r - proj(r, basis) @ basis
                b = rnorm(r)
                r = r / b
                T[k, k + 1] = b
                T[k + 1, k] = b
                ok = False
                for _ in range(10):
                    inner = proj(r, basis)
                    if not bool((inner.abs() > tol).any()):
                        ok = True
                        break
                    r = r - inner @ basis
                    r = r / rnorm(r)
                Q[k + 1] = r[0]
                if bool(b.abs() < 1e-6) or not ok:
                    break
                m = k + 2
    return Q[:m], T[:m, :m]


def tridiag_to_diag(T: torch.Tensor):
    """``lanczos_tridiag_to_diag``: eigh on the host in float64 (tiny), negative eigenvalues masked."""
    evals, evecs = torch.linalg.eigh(T.detach().to(device="cpu", dtype=torch.float64))
    mask = evals >= 0
    evecs = evecs * mask.to(evecs.dtype).unsqueeze(-2)
    evals = evals.masked_fill(~mask, 1.0)
    return evals, evecs


def root_inv_decomposition(x: B.PreparedPoints, scale, dscale, max_iter=None, init_vec_t=None, generator=None, dvec=None,
                           matvec=None, nvec=None, device=None, reduce=None, n_global=None):
    """Rt [m, ld] with Rt^T Rt ~= K_hat^-1 on the Krylov space (the ``covar_cache`` of
    ``exact_prediction_strategies.py:267-272``)."""
    max_iter = settings.max_root_decomposition_size.value() if max_iter is None else max_iter
    Q, T = lanczos_tridiag(x, scale, dscale, max_iter, init_vec_t, generator=generator, dvec=dvec, matvec=matvec,
                           nvec=nvec, device=device, reduce=reduce, n_global=n_global)
    jitter = settings.tridiagonal_jitter.value()
    Tj = T + jitter * torch.eye(T.shape[0], device=T.device, dtype=T.dtype)
    evals, evecs = tridiag_to_diag(Tj)
    w = (evecs / evals.sqrt().unsqueeze(-2)).to(device=Q.device, dtype=Q.dtype)  # V Lambda^-1/2
    return w.t() @ Q  # [m, ld]: rows = columns of Q V Lambda^-1/2
