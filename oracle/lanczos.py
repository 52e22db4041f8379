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


def lanczos_tridiag_batch(matmul_closure, max_iter: int, n: int, init_vecs: torch.Tensor, tol: float = 1e-5):
    """``lanczos_tridiag`` with ``init_vecs`` (n, b), b > 1: linear_operator runs the b columns as a trailing batch of INDEPENDENT
    recurrences that share only the product call ``matmul_closure((n, b)) -> (n, b)`` (``num_init_vecs`` in
    ``linear_operator.utils.lanczos.lanczos_tridiag``; reached through ``gpytorch.root_inv_decomposition(initial_vectors=...)``,
    gpytorch/__init__.py:190-216).  The loop ends when EVERY column's beta is below 1e-6 or a re-orthogonalisation failed.
    Returns Q (b, n, m), T (b, m, m) -- the leading dimension is the start vector, as the reference permutes it."""
    dtype = init_vecs.dtype
    b = init_vecs.shape[-1]
    num_iter = min(max_iter, n)
    Qm = torch.zeros(num_iter, n, b, dtype=dtype)
    T = torch.zeros(num_iter, num_iter, b, dtype=dtype)
    q0 = init_vecs / init_vecs.norm(2, dim=0, keepdim=True)
    Qm[0] = q0
    r = matmul_closure(q0)
    a0 = (q0 * r).sum(0)
    r = r - a0 * q0
    b0 = r.norm(2, dim=0)
    T[0, 0] = a0
    m = 1
    if num_iter > 1:
        T[0, 1] = b0
        T[1, 0] = b0
        Qm[1] = r / b0
        m = 2
        for k in range(1, num_iter):
            q_prev, q = Qm[k - 1], Qm[k]
            r = matmul_closure(q) - T[k, k - 1] * q_prev
            a = (q * r).sum(0)
            T[k, k] = a
            m = k + 1
            if k + 1 < num_iter:
                r = r - a * q
                basis = Qm[: k + 1]                                  # (k + 1, n, b)
                r = r - (basis * (basis * r).sum(1, keepdim=True)).sum(0)
                bn = r.norm(2, dim=0)
                r = r / bn
                T[k, k + 1] = bn
                T[k + 1, k] = bn
                ok = False
                for _ in range(10):
                    inner = (basis * r).sum(1)                       # (k + 1, b)
                    if not bool((inner.abs() > tol).any()):
                        ok = True
                        break
                    r = r - (basis * inner.unsqueeze(1)).sum(0)
                    r = r / r.norm(2, dim=0)
                Qm[k + 1] = r
                if int((bn.abs() > 1e-6).sum()) == 0 or not ok:
                    break
                m = k + 2
    return Qm[:m].permute(2, 1, 0).contiguous(), T[:m, :m].permute(2, 0, 1).contiguous()


def select_root_inv(matmul_closure, inv_roots: torch.Tensor, test_vectors: torch.Tensor) -> int:
    """``_postprocess_lanczos_root_inv_decomp`` (linear_operator ``_linear_operator.py``; the selection rule documented at
    gpytorch/__init__.py:190-216): inv_roots (b, n, m), test_vectors (n, c).  Every candidate solves the test vectors,
    s_i = R_i R_i^T v; the residual norms |A s_i - v|_2 are summed over the test vectors; the smallest sum wins."""
    sums = []
    for Ri in inv_roots:
        s = Ri @ (Ri.t() @ test_vectors)
        sums.append((matmul_closure(s) - test_vectors).norm(2, dim=-2).sum())
    return int(torch.stack(sums).argmin())


def root_inv_decomposition_multi(matmul_closure, n: int, max_iter: int, init_vecs: torch.Tensor, test_vectors: torch.Tensor):
    """``LinearOperator.root_inv_decomposition(initial_vectors=(n, b), test_vectors=(n, c))``: b decompositions, the best one by
    :func:`select_root_inv`.  Returns (R (n, m), index)."""
    Q, T = lanczos_tridiag_batch(matmul_closure, max_iter, n, init_vecs)
    roots = []
    for i in range(Q.shape[0]):
        evals, evecs = tridiag_to_diag(T[i])
        keep = evals > 0
        roots.append((Q[i] @ evecs[:, keep]) / evals[keep].sqrt())
    m = min(r.shape[-1] for r in roots)
    roots = torch.stack([r[:, :m] if r.shape[-1] == m else r[:, -m:] for r in roots])
    idx = select_root_inv(matmul_closure, roots, test_vectors)
    return roots[idx], idx


def block_lanczos(matmul_closure, steps: int, n: int, init_block: torch.Tensor):
    """Block Lanczos with full re-orthogonalisation (Golub & Underwood, "The block Lanczos method for computing eigenvalues", 1977;
    Golub & Van Loan, Matrix Computations 4e, section 10.3.6).  NOT a function of the reference: the checker of
    ``gpytorch_amd.lanczos.block_lanczos_steps`` (the LOVE cache on a block Krylov space).  init_block (n, b).
    Returns Q (n, steps * b) with orthonormal columns spanning [V, A V, ..., A^(steps-1) V] and T = Q^T A Q (block tridiagonal up to rounding)."""
    Qs = []
    R = init_block.clone()
    for s in range(steps):
        if Qs:
            Bq = torch.cat(Qs, 1)
            for _ in range(2):
                R = R - Bq @ (Bq.t() @ R)
        Qk, _ = torch.linalg.qr(R)
        Qs.append(Qk)
        if s + 1 < steps:
            R = matmul_closure(Qk)
    Q = torch.cat(Qs, 1)
    T = Q.t() @ matmul_closure(Q)
    return Q, 0.5 * (T + T.t())
