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
    """``lanczos_tridiag_to_diag``: eigh in float64 (on the host while the matrix is tiny, as the reference does below 32 rows; on the
    device from 256 rows on -- block decompositions of rank 400 .. 1600), negative eigenvalues masked.  Results live where they were computed."""
    T64 = T.detach().to(torch.float64)
    evals, evecs = torch.linalg.eigh(T64 if (T64.shape[-1] >= 256 and T64.is_cuda) else T64.cpu())
    mask = evals >= 0
    evecs = evecs * mask.to(evecs.dtype).unsqueeze(-2)
    evals = evals.masked_fill(~mask, 1.0)
    return evals, evecs


def lanczos_tridiag_multi(matvec, n: int, max_iter: int, init_t: torch.Tensor, tol: float = 1e-5):
    """b INDEPENDENT recurrences of :func:`_lanczos_torch` in lock-step: ``init_t`` [b, ld] holds one start vector per row and every
    step sends the b current vectors through ONE b-column operator product ``matvec([b, ld]) -> [b, ld]`` -- the reference's
    ``lanczos_tridiag(..., init_vecs=[n, b])`` (``linear_operator.utils.lanczos``, third-party; its ``num_init_vecs`` dimension is a
    batch of recurrences that share nothing but the product call; reached through ``gpytorch.root_inv_decomposition(initial_vectors=...)``,
    ``gpytorch/__init__.py:190-216``).  On the device that product is what pays: b = 5 .. 16 columns run on the matrix-pipe kernels
    (one ``kv_gramh`` launch, 45 ms for 11 columns at n = 500 000) where b single-column products are exp-bound (19 ms EACH).
    Returns Q [b, m, ld] (orthonormal rows per chain) and T [b, m, m]; same stopping rule as the reference: the loop ends when EVERY
    chain has broken down (beta < 1e-6) or a re-orthogonalisation did not converge in 10 passes."""
    b, ld = init_t.shape
    wd, dev = init_t.dtype, init_t.device
    num_iter = min(max_iter, n)
    Q = torch.zeros(b, num_iter, ld, device=dev, dtype=wd)
    T = torch.zeros(b, num_iter, num_iter, device=dev, dtype=wd)
    tiny = torch.finfo(wd).tiny

    def dot(a, c):
        return (a[:, :n] * c[:, :n]).sum(-1)

    def mv(q):
        w = matvec(q.contiguous())
        w = w if w.dtype == wd else w.to(wd)
        if ld > n:
            w[:, n:] = 0
        return w

    q0 = init_t / dot(init_t, init_t).sqrt().clamp_min(tiny).unsqueeze(-1)
    if ld > n:
        q0[:, n:] = 0
    Q[:, 0] = q0
    r = mv(q0)
    a0 = dot(q0, r)
    r = r - a0.unsqueeze(-1) * q0
    b0 = dot(r, r).sqrt()
    T[:, 0, 0] = a0
    m = 1
    if num_iter > 1:
        T[:, 0, 1] = b0
        T[:, 1, 0] = b0
        Q[:, 1] = r / b0.clamp_min(tiny).unsqueeze(-1)
        m = 2
        for k in range(1, num_iter):
            q_prev, q = Q[:, k - 1], Q[:, k]
            r = mv(q) - T[:, k, k - 1].unsqueeze(-1) * q_prev
            a = dot(q, r)
            T[:, k, k] = a
            m = k + 1
            if k + 1 < num_iter:
                r = r - a.unsqueeze(-1) * q
                basis = Q[:, : k + 1]                                          # [b, k + 1, ld]
                r = r - torch.bmm(torch.bmm(basis, r.unsqueeze(-1)).mT, basis).squeeze(1)
                bn = dot(r, r).sqrt()
                r = r / bn.clamp_min(tiny).unsqueeze(-1)
                T[:, k, k + 1] = bn
                T[:, k + 1, k] = bn
                ok = False
                for _ in range(10):
                    inner = torch.bmm(basis, r.unsqueeze(-1)).squeeze(-1)      # [b, k + 1]
                    if not bool((inner.abs() > tol).any()):                    # the one host poll of this pass
                        ok = True
                        break
                    r = r - torch.bmm(inner.unsqueeze(1), basis).squeeze(1)
                    r = r / dot(r, r).sqrt().clamp_min(tiny).unsqueeze(-1)
                Q[:, k + 1] = r
                if not bool((bn.abs() > 1e-6).any()) or not ok:
                    break
                m = k + 2
    return Q[:, :m], T[:, :m, :m]


def block_lanczos_steps(n: int, dev, steps: int, init_t: torch.Tensor):
    """BLOCK Lanczos with full re-orthogonalisation (Golub & Underwood 1977; block size b = rows of ``init_t`` [b, ld]) as a coroutine in the
    manner of :func:`lanczos_steps`: yields the probe-major block Q_s [b, ld] whose product W = A Q_s it needs next, receives W through ``send``
    and finally returns (Q [m, ld] with orthonormal rows, T [m, m] = Q A Q^T in float64, m = steps * b) as the StopIteration value.

    No counterpart in the reference -- it is what the reference's multi-vector interface becomes when the b-column product is the unit
    of cost: the span of the b lock-step recurrences of :func:`lanczos_tridiag_multi` IS the block Krylov space, and the Galerkin inverse on ALL
    of it (rank steps * b) costs the products of ONE rank-``steps`` recurrence.  For the LOVE cache the error depends on the rank of the space, not
    on how it was generated (``tests/test_block_lanczos_cpu.py``, ``tests/test_gpu_love_vs_oracle.py``), so a rank-400 cache is 50 eight-column
    products instead of 400 one-column products.
    Per step: the product, two classical Gram-Schmidt passes against every earlier block (tall-skinny GEMMs), Cholesky-QR twice on the b new
    rows (Gram matrix and factor in float64).  No host synchronisation inside the loop: a rank-deficient block (Krylov space exhausted) shows up
    in ``cholesky_ex``'s info words, read ONCE at the end; the decomposition is then cut before the first such block."""
    b, ld = init_t.shape
    wd = init_t.dtype
    m_max = steps * b
    Q = torch.zeros(m_max, ld, device=dev, dtype=wd)
    H = torch.zeros(m_max, m_max, device=dev, dtype=torch.float64)
    infos = []
    defl = (100.0 * torch.finfo(wd).eps) ** 2
    native = wd == torch.float32 and torch.device(dev).type == "cuda" and b <= 32 and not FORCE_TORCH
    if native:
        # vector work on the gpamd_block_* kernels (csrc/lanczos_kernels.hpp): float32 rows, float64 accumulation, every basis element read once per
        # pass (rocBLAS' tall-skinny float64 GEMMs -- the b x n x b Gram matrix -- cost tens of ms per call: profiles/r05_s1_love_block_timing_c2_torch.json,
        # 36 ms per step at n = 100 000 where the eight-column product itself takes 2)
        import ctypes as C

        from ._lib import check, lib

        L, st, p = lib(), B._stream(dev), B._ptr
        ws = torch.empty(max(int(L.gpamd_precond_coef_workspace_doubles(n, b, m_max)), 1), device=dev, dtype=torch.float64)
        eye = torch.eye(b, device=dev, dtype=torch.float64)

        def project(basis, k, R):
            Wc = torch.empty(b, k, device=dev, dtype=torch.float64)
            check(L.gpamd_block_project_f32(p(basis), basis.stride(0), k, p(R), R.stride(0), b, n, p(Wc), p(ws), ws.numel(), st), "block_project")
            return Wc

        def subtract(basis, k, Wc, R):
            check(L.gpamd_block_subtract_f32(p(basis), basis.stride(0), k, p(Wc), p(R), R.stride(0), b, n, st), "block_subtract")

        def chol_qr(R):
            G = project(R, b, R)                                           # R R^T (float64 accumulation)
            G = 0.5 * (G + G.t())
            Lc, inf = torch.linalg.cholesky_ex(G)
            Minv = torch.linalg.solve_triangular(Lc, eye, upper=False).contiguous()
            check(L.gpamd_block_transform_f32(p(Minv), p(R), R.stride(0), b, n, st), "block_transform")
            return inf, G.diagonal(), Lc.diagonal() ** 2
    else:
        def project(basis, k, R):
            return (R[:, :n] @ basis[:k, :n].t()).to(torch.float64)

        def subtract(basis, k, Wc, R):
            R[:, :n] -= Wc.to(wd) @ basis[:k, :n]

        def chol_qr(R):
            R64 = R[:, :n].to(torch.float64)
            G = R64 @ R64.t()
            Lc, inf = torch.linalg.cholesky_ex(G)
            R[:, :n] = torch.linalg.solve_triangular(Lc, R64, upper=False).to(wd)
            return inf, G.diagonal(), Lc.diagonal() ** 2

    def orthonormalise(R, k):
        """Rows of R (modified in place) made orthonormal and orthogonal to Q[:k]; two passes ("twice is enough").  Returns the projection
        coefficients of the FIRST pass ([b, k] float64: the column block of Q A Q^T when R = A Q_s) or None."""
        info, first = None, None
        for it in range(2):
            if k:
                Wc = project(Q, k, R)
                subtract(Q, k, Wc, R)
                first = Wc if it == 0 else first
            inf, g_diag, l_diag2 = chol_qr(R)
            if it == 0:
                # deflation: a row whose part orthogonal to everything before it (earlier blocks AND earlier rows of this block: the Cholesky pivot) is
                # rounding noise relative to the row itself (|w_c|^2 = |coefficients|^2 + |residual|^2) -- the block Krylov space is exhausted
                wn2 = g_diag if first is None else g_diag + (first * first).sum(-1)
                inf = torch.maximum(inf, (l_diag2 < defl * wn2).any().to(inf.dtype))
            info = inf if info is None else torch.maximum(info, inf)
        infos.append(info)
        return first

    R0 = init_t.clone()
    if ld > n:
        R0[:, n:] = 0
    orthonormalise(R0, 0)
    Q[:b] = R0
    for s in range(steps):
        k = (s + 1) * b
        W = yield Q[k - b : k]
        R = torch.zeros(b, ld, device=dev, dtype=wd)
        R[:, :n] = W[:, :n]
        if s + 1 == steps:
            H[:k, k - b : k] = project(Q, k, R).t()
            break
        H[:k, k - b : k] = orthonormalise(R, k).t()
        Q[k : k + b] = R
    bad = [i for i, v in enumerate(torch.stack(infos).reshape(-1).tolist()) if v]       # the one host synchronisation
    m = m_max if not bad else bad[0] * b
    if m == 0:
        raise RuntimeError("block Lanczos: the start block is rank deficient")
    if m < m_max:
        # (advisor finding, round 5) a shortened decomposition is said out loud: the LOVE cache built on it has a lower rank than
        # settings.max_root_decomposition_size asked for, and only the variance error would show it otherwise
        import warnings

        from .linear_cg import NumericalWarning

        warnings.warn(f"block Lanczos stopped at rank {m} of the {m_max} requested (block {bad[0]} of {steps}: rank-deficient or ill-conditioned in "
                      "float32 -- the block Krylov space is exhausted to rounding); decompositions built on it have that rank.", NumericalWarning)
    Hm = H[:m, :m]
    T = torch.triu(Hm) + torch.triu(Hm, 1).t()
    return Q[:m], T


def block_lanczos(matvec, n: int, dev, steps: int, init_t: torch.Tensor):
    """Drives :func:`block_lanczos_steps` with ``matvec([b, ld]) -> [b, ld]``."""
    gen = block_lanczos_steps(n, dev, steps, init_t)
    try:
        q = next(gen)
        while True:
            q = gen.send(matvec(q))
    except StopIteration as done:
        return done.value


def block_size_for(n: int, rank: int) -> int:
    """Block size of the LOVE / root decompositions: ``settings.lanczos_block_size`` ("auto": 8 rows per product once a product fills
    the chip -- n >= 16 384 -- AND the requested rank is at least ``auto_min_rank`` = 200; otherwise the reference's single-vector recurrence).
    The rank threshold is a fidelity rule, not a speed rule: a block Krylov space of dimension k holds less of the spectrum's ends than the
    single-vector space of the same dimension while k is small -- variance error over noise at n = 20 000 (oracle, float64;
    profiles/r05_s1_love_vs_oracle_n20000.json): rank 50 3.67 against 2.77, rank 100 0.750 against 0.573 (C2 at rank 96 / 100: 1.01 against 0.63),
    rank 200 0.02602 against 0.02625, rank 400 equal -- so below rank 200 the default stays the reference's algorithm at the reference's accuracy,
    from 200 on the two are indistinguishable and the block form is 3-5 x faster."""
    v = settings.lanczos_block_size.value()
    if v == "auto":
        if n < settings.lanczos_block_size.auto_min_size or rank < settings.lanczos_block_size.auto_min_rank:
            return 1
        return 32 if n >= settings.lanczos_block_size.auto_wide_size else 8     # (round 6: <= 32 columns cost one generation-bound launch at large n)
    return max(1, min(int(v), rank))


def root_from_tridiag(Q: torch.Tensor, T: torch.Tensor, inverse: bool = True) -> torch.Tensor:
    """Rows of (V Lambda^-1/2)^T Q (``inverse``) or (V Lambda^1/2)^T Q for T = V Lambda V^T, eigendecomposition on the host in float64
    with ``settings.tridiagonal_jitter`` on the diagonal and negative eigenvalues masked (``lanczos_tridiag_to_diag``)."""
    Tj = T + settings.tridiagonal_jitter.value() * torch.eye(T.shape[-1], device=T.device, dtype=T.dtype)
    evals, evecs = tridiag_to_diag(Tj)
    sc = evals.sqrt().unsqueeze(-2)
    w = (evecs / sc if inverse else evecs * sc).to(device=Q.device, dtype=Q.dtype)
    return w.mT @ Q


def select_by_test_vectors(roots_t: torch.Tensor, test_t: torch.Tensor, matvec, n: int) -> int:
    """``_postprocess_lanczos_root_inv_decomp`` (linear_operator, third-party; documented at ``gpytorch/__init__.py:190-216``: "the best
    initialization vector (determined by ``test_vectors``) will be chosen"): every candidate root R_i (rows of ``roots_t[i]``, R_i^T R_i ~= A^-1)
    solves the test vectors, ONE product with all b * c solves gives the residuals |A s - v|_2, summed over the test vectors; the index of the
    smallest sum is returned."""
    b, _, ld = roots_t.shape
    c = test_t.shape[0]
    test_t = test_t.to(roots_t.dtype)
    coef = torch.einsum("cl,bml->bcm", test_t[:, :n], roots_t[:, :, :n])
    sol = torch.einsum("bcm,bml->bcl", coef, roots_t)
    prod = matvec(sol.reshape(b * c, ld).contiguous()).reshape(b, c, ld)
    res = (prod[:, :, :n] - test_t[:, :n].unsqueeze(0)).norm(dim=-1).sum(-1)
    return int(res.argmin())


def root_inv_decomposition(x: B.PreparedPoints, scale, dscale, max_iter=None, init_vec_t=None, generator=None, dvec=None,
                           matvec=None, nvec=None, device=None, reduce=None, n_global=None, test_vec_t=None, block=None, dtype=None):
    """Rt [m, ld] with Rt^T Rt ~= K_hat^-1 on the Krylov space (the ``covar_cache`` of
    ``exact_prediction_strategies.py:267-272``).

    ``init_vec_t`` with b > 1 rows is the reference's multi-vector form (``gpytorch.root_inv_decomposition(initial_vectors, test_vectors)``,
    ``gpytorch/__init__.py:190-216``): b recurrences in lock-step, the one whose root solves ``test_vec_t`` best is returned.  Without start
    vectors (or with one) ``block`` (default :func:`block_size_for`) > 1 builds the cache on a block Krylov space instead (:func:`block_lanczos_steps`)."""
    max_iter = settings.max_root_decomposition_size.value() if max_iter is None else max_iter
    n = x.n if nvec is None else nvec
    dev = x.xp.device if device is None else device

    def mv(q_rows):
        if matvec is not None:
            return matvec(q_rows)
        return B.kv(x, x, q_rows, scale=scale, dscale=dscale, vd=q_rows if dscale is not None else None, dvec=dvec)

    if init_vec_t is not None and init_vec_t.shape[0] > 1:
        if reduce is not None:
            raise NotImplementedError("root_inv_decomposition: several initial vectors on row-sharded vectors")
        if test_vec_t is None:
            raise ValueError("root_inv_decomposition: several initial_vectors need test_vectors to choose between their decompositions "
                             "(gpytorch/__init__.py:190-216)")
        Qb, Tb = lanczos_tridiag_multi(mv, n, max_iter, init_vec_t)
        roots = torch.stack([root_from_tridiag(Qb[i], Tb[i]) for i in range(Qb.shape[0])])
        return roots[select_by_test_vectors(roots, test_vec_t, mv, n)]
    block = block_size_for(n if n_global is None else n_global, max_iter) if block is None else block
    if block > 1 and reduce is None and init_vec_t is None:
        ld = B.round_up(n, 4)
        wd = dtype if dtype is not None else (x.dtype if x is not None else torch.float32)
        init = torch.zeros(block, ld, device=dev, dtype=wd)
        init[:, :n] = torch.randn(block, n, device=dev, generator=generator, dtype=wd)
        Q, T = block_lanczos(mv, n, dev, max(1, min(max_iter, n) // block), init)
        return root_from_tridiag(Q, T)
    Q, T = lanczos_tridiag(x, scale, dscale, max_iter, init_vec_t, generator=generator, dvec=dvec, matvec=matvec,
                           nvec=nvec, device=device, reduce=reduce, n_global=n_global)
    return root_from_tridiag(Q, T)
