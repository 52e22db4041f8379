"""The oracle's iterative algorithms against INDEPENDENT third-party implementations of the same mathematics (SciPy / LAPACK).

`oracle/linear_cg.py`, `oracle/lanczos.py` and `oracle/pivoted_cholesky.py` restate `linear_operator` (third-party, not vendored under
/root/reference, no golden vectors exist: "parity unpinned at iteration level").  What CAN be pinned without the
package is that they are the textbook recurrences the package documents, step for step:
  * mBCG column = conjugate gradients: the iterate after k steps equals scipy.sparse.linalg.cg's after k steps (x0 = 0), with and
    without a preconditioner;
  * the tridiagonal matrix mBCG assembles from its alpha / beta history is the Lanczos tridiagonalisation started at b / |b|
    (the CG-Lanczos equivalence the SLQ log-det rests on): equal to an independent Lanczos with full re-orthogonalisation;
  * the oracle's own Lanczos returns the same T and an orthonormal Q with Q^T A Q = T;
  * pivoted Cholesky = LAPACK's dpstrf (diagonal pivoting): same pivots, same factor, column by column;
  * stochastic Lanczos quadrature of the exact T (k = n) is exact: e_1^T log(T) e_1 |b|^2 == b^T log(A) b.
"""
import numpy as np
import pytest
import scipy.linalg
import scipy.linalg.lapack
import scipy.sparse.linalg as spla
import torch

from oracle import kernels as OK
from oracle.lanczos import lanczos_tridiag
from oracle.linear_cg import linear_cg
from oracle.pivoted_cholesky import build_preconditioner, pivoted_cholesky


def _system(n=120, seed=0, ls=0.3, noise=1e-2):
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(n, 2, generator=g, dtype=torch.float64)
    A = OK.rbf(X, X, ls, direct=True) + noise * torch.eye(n, dtype=torch.float64)
    b = torch.randn(n, 1, generator=g, dtype=torch.float64)
    return X, A, b


def _scipy_cg_iterate(A, b, k, M=None):
    """x_k of SciPy's CG (x0 = 0): stop it after exactly k iterations."""
    x, _ = spla.cg(A.numpy(), b[:, 0].numpy(), x0=np.zeros(A.shape[0]), rtol=0.0, atol=0.0, maxiter=k, M=M)
    return torch.from_numpy(x)


@pytest.mark.parametrize("k", [1, 2, 5, 17, 40])
def test_mbcg_iterates_are_scipy_cg_iterates(k):
    # a well-conditioned system: on an ill-conditioned one two correct CG codes drift apart in their late iterates (rounding
    # errors are amplified by the loss of orthogonality), which says nothing about either
    _, A, b = _system(noise=0.5)
    x = linear_cg(lambda v: A @ v, b, tolerance=0.0, max_iter=k, eps=1e-300, stop_updating_after=1e-300)[:, 0]
    ref = _scipy_cg_iterate(A, b, k)
    assert float((x - ref).norm() / ref.norm()) < 1e-8


@pytest.mark.parametrize("k", [1, 3, 9, 25])
def test_preconditioned_mbcg_iterates_are_scipy_pcg_iterates(k):
    X, A, b = _system(noise=0.5)
    K = A - 0.5 * torch.eye(A.shape[0], dtype=torch.float64)
    L = pivoted_cholesky(K.diagonal(), lambda p: K[p], rank=8, error_tol=1e-12)
    pre, _, _ = build_preconditioner(L, 0.5)
    x = linear_cg(lambda v: A @ v, b, tolerance=0.0, max_iter=k, eps=1e-300, stop_updating_after=1e-300, preconditioner=pre)[:, 0]
    P = L @ L.t() + 0.5 * torch.eye(A.shape[0], dtype=torch.float64)
    M = spla.LinearOperator(A.shape, matvec=lambda v: np.linalg.solve(P.numpy(), v))
    ref = _scipy_cg_iterate(A, b, k, M=M)
    assert float((x - ref).norm() / ref.norm()) < 1e-8


def _lanczos_full_reorth(A, q0, m):
    """Independent Lanczos (numpy, two passes of Gram-Schmidt against all previous vectors)."""
    n = A.shape[0]
    Q = np.zeros((n, m))
    alpha, beta = np.zeros(m), np.zeros(m - 1)
    q = q0 / np.linalg.norm(q0)
    for j in range(m):
        Q[:, j] = q
        w = A @ q
        alpha[j] = q @ w
        for _ in range(2):
            w = w - Q[:, : j + 1] @ (Q[:, : j + 1].T @ w)
        if j + 1 < m:
            beta[j] = np.linalg.norm(w)
            q = w / beta[j]
    return Q, alpha, beta


def test_mbcg_tridiagonal_is_the_lanczos_tridiagonal():
    _, A, b = _system(n=90, seed=3)
    m = 12
    _, Tm = linear_cg(lambda v: A @ v, b, n_tridiag=1, tolerance=0.0, max_iter=m, max_tridiag_iter=m, eps=1e-300,
                      stop_updating_after=1e-300)
    T = Tm[0]
    _, alpha, beta = _lanczos_full_reorth(A.numpy(), b[:, 0].numpy(), m)
    Tref = np.diag(alpha) + np.diag(beta, 1) + np.diag(beta, -1)
    # mBCG's T carries the off-diagonals with the sign convention beta >= 0 as well; compare entries and Ritz values
    np.testing.assert_allclose(np.abs(T.numpy()), np.abs(Tref), rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(np.linalg.eigvalsh(T.numpy()), np.linalg.eigvalsh(Tref), rtol=1e-6, atol=1e-10)


def test_oracle_lanczos_matches_independent_lanczos():
    _, A, b = _system(n=80, seed=5)
    m = 15
    Q, T = lanczos_tridiag(lambda v: A @ v, m, A.shape[0], b.clone(), tol=1e-12)
    Qr, alpha, beta = _lanczos_full_reorth(A.numpy(), b[:, 0].numpy(), m)
    Tref = np.diag(alpha) + np.diag(beta, 1) + np.diag(beta, -1)
    T2 = T.squeeze().numpy()
    np.testing.assert_allclose(np.abs(T2), np.abs(Tref), rtol=1e-7, atol=1e-10)
    Q2 = Q.squeeze().numpy()
    np.testing.assert_allclose(Q2.T @ Q2, np.eye(m), atol=1e-9)
    np.testing.assert_allclose(Q2.T @ A.numpy() @ Q2, T2, atol=1e-8)
    np.testing.assert_allclose(np.abs(Q2.T @ Qr), np.eye(m), atol=1e-7)     # same Krylov basis up to signs


def test_pivoted_cholesky_is_lapack_dpstrf():
    X, A, _ = _system(n=70, seed=7, ls=0.4, noise=0.0)
    K = A + 1e-10 * torch.eye(70, dtype=torch.float64)
    rank = 20
    L, piv = pivoted_cholesky(K.diagonal(), lambda p: K[p], rank=rank, error_tol=0.0, return_pivots=True)
    c, lpiv, lrank, linfo = scipy.linalg.lapack.dpstrf(K.numpy(), lower=1, tol=1e-300)
    assert linfo in (0, 1) and lrank >= rank
    lpiv = lpiv - 1                                           # LAPACK pivots are 1-based: P^T K P = C C^T
    # same pivot sequence (distinct diagonal maxima on this data: no ties) ...
    assert [int(p) for p in piv[:rank]] == [int(p) for p in lpiv[:rank]]
    # ... and the same factor: row i of LAPACK's C belongs to original row lpiv[i]
    C = np.tril(c)[:, :rank]
    Lref = np.zeros_like(C)
    Lref[lpiv, :] = C
    np.testing.assert_allclose(L.numpy(), Lref, rtol=1e-8, atol=1e-10)


def test_lanczos_quadrature_of_the_full_tridiagonal_is_exact():
    _, A, b = _system(n=40, seed=9, noise=0.1)
    n = A.shape[0]
    Q, T = lanczos_tridiag(lambda v: A @ v, n, n, b.clone(), tol=1e-14)
    T2 = T.squeeze().numpy()
    ev, U = np.linalg.eigh(T2)
    quad = float((U[0] ** 2 * np.log(ev)).sum()) * float(b.norm() ** 2)
    ref = float(b[:, 0].numpy() @ scipy.linalg.logm(A.numpy()).real @ b[:, 0].numpy())
    assert abs(quad - ref) < 1e-6 * abs(ref)
