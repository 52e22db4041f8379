"""Lanczos tridiagonalisation with full re-orthogonalisation and the root / root-inverse
decompositions built on it (LOVE predictive-variance cache).

Mirrors ``linear_operator.utils.lanczos.lanczos_tridiag`` / ``lanczos_tridiag_to_diag`` and
``LinearOperator.root_inv_decomposition`` (third-party; restated in ``oracle/lanczos.py``,
SURVEY.md A.7).  Reference call sites: ``gpytorch/models/exact_prediction_strategies.py:202,
234-238,271``.  Each step is one fused K_hat*q (t = 1 -> the VALU kernel: generation-bound) plus
O(n k) re-orthogonalisation on probe-major rows (rocBLAS GEMV through torch).
"""
from __future__ import annotations

import torch

from . import backend as B
from . import settings


def lanczos_tridiag(x: B.PreparedPoints, scale, dscale, max_iter: int, init_vec_t: torch.Tensor | None = None,
                    tol: float = 1e-5, generator=None, dvec=None, matvec=None, nvec=None, device=None, reduce=None, n_global=None):
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
