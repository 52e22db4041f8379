"""Oracle: Lanczos tridiagonalisation with full re-orthogonalisation and the root
(inverse) decompositions built on it.  Test infrastructure only.

Restates the published algorithm of ``linear_operator.utils.lanczos.lanczos_tridiag`` and
``lanczos_tridiag_to_diag`` (linear_operator v0.6.x, third-party, not vendored).
Reference call sites: ``gpytorch/models/exact_prediction_strategies.py:202,234-238,271``
(``root_inv_decomposition`` for the LOVE ``covar_cache``).
Iteration-level parity with the reference is UNPINNED; pinned via Q^T Q = I,
Q T Q^T ~= A on the Krylov space and predictive variances vs dense Cholesky, and against an independent Lanczos
(same T, same Krylov basis up to signs; exact quadrature b^T log(A) b for k = n: tests/test_oracle_independent_cpu.py).
"""
from __future__ import annotations

import torch


def lanczos_tridiag(matmul_closure, max_iter: int, n: int, init_vec: torch.Tensor, tol: float = 1e-5):
    """Returns Q (n, m) and T (m, m).  init_vec: (n, 1)."""
    dtype = init_vec.dtype
    num_iter = min(max_iter, n)
    Qm = torch.zeros(num_iter, n, dtype=dtype)
    T = torch.zeros(num_iter, num_iter, dtype=dtype)

    q0 = (init_vec / init_vec.norm(2, dim=-2, keepdim=True)).reshape(-1)
    Qm[0] = q0
    r = matmul_closure(q0.unsqueeze(-1)).reshape(-1)
    a0 = q0.dot(r)
    r = r - a0 * q0
    b0 = r.norm()
    T[0, 0] = a0
    m = 1
    if num_iter > 1:
        T[0, 1] = b0
        T[1, 0] = b0
        Qm[1] = r / b0
        m = 2
        for k in range(1, num_iter):
            q_prev, q = Qm[k - 1], Qm[k]
            b_prev = T[k, k - 1]
            r = matmul_closure(q.unsqueeze(-1)).reshape(-1) - b_prev * q_prev
            a = q.dot(r)
            T[k, k] = a
            m = k + 1
            if k + 1 < num_iter:
                r = r - a * q
                basis = Qm[: k + 1]
                r = r - basis.t() @ (basis @ r)
                b = r.norm()
                r = r / b
                T[k, k + 1] = b
                T[k + 1, k] = b
                ok = False
                for _ in range(10):
                    inner = basis @ r
                    if not bool((inner.abs() > tol).any()):
                        ok = True
                        break
                    r = r - basis.t() @ inner
                    r = r / r.norm()
                Qm[k + 1] = r
                if bool(b.abs() < 1e-6) or not ok:
                    break
                m = k + 2
    return Qm[:m].t().contiguous(), T[:m, :m].contiguous()


def tridiag_to_diag(T: torch.Tensor):
    """lanczos_tridiag_to_diag: eigh of (batched) T."""
    evals, evecs = torch.linalg.eigh(T.to(torch.float64))
    return evals.to(T.dtype), evecs.to(T.dtype)


def root_inv_decomposition(matmul_closure, n: int, max_iter: int, init_vec: torch.Tensor):
    """Lanczos root-inverse: returns R (n, m) with R R^T ~= A^-1 on the Krylov space."""
    Q, T = lanczos_tridiag(matmul_closure, max_iter, n, init_vec)
    evals, evecs = tridiag_to_diag(T)
    keep = evals > 0
    Qe = Q @ evecs[:, keep]
    return Qe / evals[keep].sqrt()
