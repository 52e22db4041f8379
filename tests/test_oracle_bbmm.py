"""CPU: pin the oracle's BBMM restatements (mBCG, SLQ, pivoted Cholesky, preconditioner, Lanczos)
through their RESULTS against dense float64 Cholesky -- the deterministic ground truth the
reference's own tests use (test/lazy/test_lazy_evaluated_kernel_tensor.py:84-105: rtol 0.02;
test/distributions/test_multivariate_normal.py:219-237: 1e-2).  Iteration-level parity with
linear_operator is unpinned (no golden vectors exist for it)."""
import math

import pytest
import torch

from oracle import exact_gp as OG
from oracle import kernels as OK
from oracle import lanczos as OL
from oracle import linear_cg as OCG
from oracle import pivoted_cholesky as OPC
from oracle import slq as OS
from tests.util import make_data


@pytest.mark.parametrize("kind,d,ls", [("rbf", 3, 0.25), ("matern52", 10, 0.8), ("matern12", 2, 0.5)])
def test_cg_solve_matches_cholesky(kind, d, ls):
    n = 400
    X, y = make_data(n, d)
    rhs = torch.randn(n, 6, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    mm = OG.make_matmul(kind, X, ls, 1.0, 0.1)
    sol = OCG.linear_cg(mm, rhs, tolerance=1e-4, max_iter=400)
    ref, _ = OG.dense_solve_logdet(kind, X, rhs, ls, 1.0, 0.1)
    assert torch.allclose(sol, ref, rtol=0.02, atol=1e-3)
    assert (sol - ref).abs().max() / ref.abs().max() < 1e-3


def test_cg_chunked_equals_dense_matmul():
    n = 300
    X, y = make_data(n, 3)
    V = torch.randn(n, 4, dtype=torch.float64)
    for kind in ("rbf", "matern32"):
        a = OG.make_matmul(kind, X, 0.3, 1.2, 0.1, dense=True)(V)
        b = OG.make_matmul(kind, X, 0.3, 1.2, 0.1, dense=False, chunk=64)(V)
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-8)


def test_cg_stopping_rule_and_tridiag_shape():
    n = 300
    X, y = make_data(n, 3)
    mm = OG.make_matmul("rbf", X, 0.25, 1.0, 0.1)
    rhs = torch.randn(n, 5, dtype=torch.float64)
    _, T, info = OCG.linear_cg(mm, rhs, n_tridiag=3, tolerance=1.0, return_info=True)
    assert info["iters"] == 21 and T.shape == (3, 20, 20)
    _, info = OCG.linear_cg(mm, rhs, tolerance=1.0, return_info=True)
    assert info["iters"] == 11
    z = torch.zeros(n, 2, dtype=torch.float64)
    z[:, 0] = y
    sol = OCG.linear_cg(mm, z, tolerance=1e-8, max_iter=300)
    assert torch.equal(sol[:, 1], torch.zeros(n, dtype=torch.float64))


def test_pivoted_cholesky_properties():
    n = 250
    X, _ = make_data(n, 2)
    K = OK.kernel_matrix("rbf", X, X, 0.4, 1.3, x1_eq_x2=True)
    L, piv = OPC.pivoted_cholesky(K.diagonal().clone(), lambda p: K[p], 60, 1e-6, return_pivots=True)
    m = L.shape[1]
    # exact on the pivot rows/cols, PSD residual, error shrinks with rank
    res = K - L @ L.t()
    assert res[piv[:m]].abs().max() < 1e-9
    assert torch.linalg.eigvalsh(res).min() > -1e-9
    L10 = OPC.pivoted_cholesky(K.diagonal().clone(), lambda p: K[p], 10, 0.0)
    assert (K - L10 @ L10.t()).diagonal().sum() > res.diagonal().sum()
    # against scipy-free reference: full-rank pivoted Cholesky reproduces K
    Lf = OPC.pivoted_cholesky(K.diagonal().clone() + 1e-6, lambda p: K[p] + 1e-6 * torch.eye(n, dtype=K.dtype)[p], n, 0.0)
    assert torch.allclose(Lf @ Lf.t(), K + 1e-6 * torch.eye(n, dtype=K.dtype), atol=1e-7)


def test_preconditioner_is_exact_inverse_of_P():
    n, k = 300, 20
    X, _ = make_data(n, 3)
    K = OK.kernel_matrix("rbf", X, X, 0.25, 1.0, x1_eq_x2=True)
    L = OPC.pivoted_cholesky(K.diagonal().clone(), lambda p: K[p], k, 1e-9)
    apply, logdet, Q1 = OPC.build_preconditioner(L, 0.1)
    P = L @ L.t() + 0.1 * torch.eye(n, dtype=K.dtype)
    V = torch.randn(n, 3, dtype=K.dtype)
    assert torch.allclose(apply(V), torch.linalg.solve(P, V), rtol=1e-8, atol=1e-10)
    assert abs(float(logdet) - float(torch.linalg.slogdet(P)[1])) < 1e-8


@pytest.mark.parametrize("rank", [0, 15])
def test_slq_logdet_and_mll_vs_cholesky(rank):
    n = 500
    X, y = make_data(n, 3)
    exact = OG.dense_mll("rbf", X, y, 0.25, 1.0, 0.1)
    est = OG.bbmm_mll("rbf", X, y, 0.25, 1.0, 0.1, num_probes=64, precond_rank=rank, min_precond_size=100, cg_tol=1e-5)
    assert abs(float(est) - float(exact)) < 0.02  # per-datum MLL; SLQ with 64 probes is a ~1% estimator of log|K|
    # the full-Krylov limit of SLQ with a complete probe basis is exact: T from n-step CG on e_i
    m = 40
    Xs, _ = make_data(m, 2, seed=3)
    Kh = OK.kernel_matrix("rbf", Xs, Xs, 0.5, 1.0, x1_eq_x2=True) + 0.5 * torch.eye(m, dtype=torch.float64)
    Zb = torch.eye(m, dtype=torch.float64)
    _, T = OCG.linear_cg(lambda v: Kh @ v, Zb, n_tridiag=m, tolerance=0.0, max_iter=m, max_tridiag_iter=m)
    ld = OS.slq_logdet(T, m)
    assert abs(float(ld) - float(torch.linalg.slogdet(Kh)[1])) < 1e-6 * m


def test_mll_gradients_formula():
    """A.6 backward with a complete probe basis equals the float64 autograd gradient of the dense MLL."""
    n = 60
    X, y = make_data(n, 2)
    Z = torch.eye(n, dtype=torch.float64) * math.sqrt(n)  # z z^T sums to n I -> exact trace
    _, aux = OG.bbmm_mll("rbf", X, y, 0.4, 1.2, 0.2, precond_rank=0, cg_tol=0.0, max_cg_iter=200, probes=Z, return_aux=True)
    g = OG.bbmm_mll_grads("rbf", X, aux, 0.4, 1.2, 0.2)
    _, gref = OG.dense_mll_and_grads("rbf", X, y, 0.4, 1.2, 0.2)
    for a, b in zip(g, gref):
        assert abs(float(a) - float(b)) < 1e-6 * max(1.0, abs(float(b)))


def test_lanczos_and_love_variance():
    n = 300
    X, y = make_data(n, 3)
    Kh = OK.kernel_matrix("rbf", X, X, 0.25, 1.0, x1_eq_x2=True) + 0.1 * torch.eye(n, dtype=torch.float64)
    init = torch.randn(n, 1, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    Q, T = OL.lanczos_tridiag(lambda v: Kh @ v, 50, n, init)
    assert torch.allclose(Q.t() @ Q, torch.eye(Q.shape[1], dtype=torch.float64), atol=1e-8)
    assert torch.allclose(Q.t() @ Kh @ Q, T, atol=1e-7)
    Xs, _ = make_data(40, 3, seed=9)
    mu, var = OG.dense_posterior("rbf", X, y, Xs, 0.25, 1.0, 0.1)
    mu2, var2 = OG.bbmm_posterior("rbf", X, y, Xs, 0.25, 1.0, 0.1, eval_cg_tol=1e-6, min_precond_size=10**9, love_rank=n)
    assert torch.allclose(mu, mu2, atol=1e-5)
    assert ((var - var2).abs() / var).max() < 0.05  # reference: within 5 % (test_simple_gp_regression.py:440-442)
    mu3, var3 = OG.bbmm_posterior("rbf", X, y, Xs, 0.25, 1.0, 0.1, eval_cg_tol=1e-6, min_precond_size=10**9, fast_pred_var=False)
    assert torch.allclose(var, var3, rtol=1e-4)
