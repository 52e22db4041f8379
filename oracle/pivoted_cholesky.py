"""Oracle: pivoted Cholesky, the low-rank-plus-diagonal preconditioner and probe vectors.
Test infrastructure only.

Restates the published algorithms of linear_operator v0.6.x (third-party, not vendored):
  * ``functions/_pivoted_cholesky.py``   -- wrapper in the reference:
    ``gpytorch/__init__.py:146-173``; also used by ``likelihoods/multitask_gaussian_likelihood.py:285``
  * ``operators/added_diag_linear_operator.py::_preconditioner/_init_cache``
  * ``LinearOperator._probe_vectors_and_norms``
Iteration-level parity with the reference is UNPINNED; results pinned by properties
(L L^T -> K as rank grows; P^-1 exact vs dense inverse; logdet(P) vs dense slogdet) and, step for step, against LAPACK's
diagonally pivoted Cholesky dpstrf (same pivots, same factor columns: tests/test_oracle_independent_cpu.py).
"""
from __future__ import annotations

import math

import torch


def pivoted_cholesky(diag: torch.Tensor, row_fn, rank: int, error_tol: float = 1e-3, return_pivots: bool = False,
                     forced_pivots=None, return_gaps: bool = False):
    """Greedy partial Cholesky of a PSD matrix given its diagonal and a row oracle.

    ``row_fn(p)`` returns row ``K[p, :]`` (n,).  Returns L of shape (n, m), m <= rank.
    Ties in the pivot search resolve to the lowest position in the current permutation
    (torch.max semantics on CPU), exactly as the sequential reference does.

    ``forced_pivots`` (test aid): follow a given pivot sequence instead of the arg-max and, with
    ``return_gaps``, report for each step ``max(live d) - d[forced pivot]`` so a test can assert the
    device's choice was an arg-max up to rounding (exact ties are legion in float32).
    """
    d = diag.clone()
    n = d.shape[-1]
    max_iter = min(rank, n)
    L = torch.zeros(max_iter, n, dtype=d.dtype)
    orig_error = d.max()
    errors = d.abs().sum() / orig_error
    perm = torch.arange(n)
    m = 0
    gaps = []
    while m == 0 or (m < max_iter and errors > error_tol):
        if forced_pivots is not None and m >= len(forced_pivots):
            break
        vals = d[perm[m:]]
        j = int(torch.argmax(vals)) + m
        if forced_pivots is not None:
            jf = int((perm == int(forced_pivots[m])).nonzero()[0])
            assert jf >= m, "forced pivot was already used"
            gaps.append(float(vals[j - m] - vals[jf - m]))
            j = jf
        maxval = vals[j - m]
        tmp = perm[m].clone()
        perm[m] = perm[j]
        perm[j] = tmp
        p = int(perm[m])
        L[m, p] = maxval.sqrt()
        row = row_fn(p)
        if m + 1 < n:
            idx = perm[m + 1 :]
            v = row[idx].clone()
            if m > 0:
                v -= (L[:m, p].unsqueeze(-1) * L[:m, idx]).sum(0)
            v /= L[m, p]
            L[m, idx] = v
            d[idx] = d[idx] - v.pow(2)
            errors = d[idx].abs().sum() / orig_error
        m += 1
    Lt = L[:m].t().contiguous()
    if return_gaps:
        return Lt, perm, gaps
    return (Lt, perm) if return_pivots else Lt


def build_preconditioner(L: torch.Tensor, sigma2: float):
    """A.4: QR of [L; sqrt(s2) I_k]; returns (apply_closure, logdet_P, Q1)."""
    n, k = L.shape
    stacked = torch.cat([L, math.sqrt(sigma2) * torch.eye(k, dtype=L.dtype)], dim=-2)
    Q, R = torch.linalg.qr(stacked)
    Q1 = Q[:n]

    def apply(V):
        return (V - Q1 @ (Q1.t() @ V)) / sigma2

    logdet = 2.0 * R.diagonal().abs().log().sum() + (n - k) * math.log(sigma2)
    return apply, logdet, Q1


def probe_vectors(n: int, t: int, L: torch.Tensor | None, sigma2: float, generator: torch.Generator, dtype=torch.float64):
    """A.5: Rademacher probes without a preconditioner, N(0, L L^T + s2 I) samples with one.
    Returns (Z normalised per column, norms (1, t))."""
    if L is None:
        Z = torch.randint(0, 2, (n, t), generator=generator).to(dtype) * 2 - 1
    else:
        k = L.shape[-1]
        e1 = torch.randn(k, t, generator=generator, dtype=dtype)
        e2 = torch.randn(n, t, generator=generator, dtype=dtype)
        Z = L @ e1 + math.sqrt(sigma2) * e2
    norms = Z.norm(2, dim=-2, keepdim=True)
    return Z / norms, norms
