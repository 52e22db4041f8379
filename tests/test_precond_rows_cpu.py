"""CPU: the host-side (pure torch) parts of the structured-operator preconditioner -- the row-callback pivoted Cholesky and the
non-constant-diagonal preconditioner P = L L^T + D -- against the oracle's pivoted Cholesky (== LAPACK dpstrf,
tests/test_oracle_independent_cpu.py) and dense float64 algebra.  Reference: ``AddedDiagLinearOperator._preconditioner`` /
``_init_cache_for_non_constant_diag`` (third-party linear_operator), reached for every summand structure through
``gpytorch/settings.py:6-31`` (max_preconditioner_size)."""
import torch

from gpytorch_amd.bbmm import pivoted_cholesky_rows, preconditioner_from_factor
from oracle import kernels as OK
from oracle import pivoted_cholesky as OPC


def _kmat(n=300, d=2, seed=0):
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(n, d, generator=g, dtype=torch.float64)
    return 1.3 * OK.rbf(X, X, 0.3, x1_eq_x2=True) + 0.4 * OK.matern(X, X, 0.7, 1.5, x1_eq_x2=True)


def test_pivoted_cholesky_rows_matches_the_oracle():
    K = _kmat()
    rank = 25
    lt = pivoted_cholesky_rows(lambda p: K[p], K.diagonal().clone(), rank, 1e-12)
    L_ref = OPC.pivoted_cholesky(K.diagonal().clone(), lambda p: K[p], rank, error_tol=1e-12)          # [n, k]
    k = L_ref.shape[-1]
    assert k == rank
    assert torch.allclose(lt.t(), L_ref, atol=1e-9)
    # the tolerance rule: once the remaining trace is below tol * max diag, further steps write zero rows
    lt2 = pivoted_cholesky_rows(lambda p: K[p], K.diagonal().clone(), 200, 5.0)
    L2 = OPC.pivoted_cholesky(K.diagonal().clone(), lambda p: K[p], 200, error_tol=5.0)
    k2 = L2.shape[-1]
    assert 5 < k2 < 200
    assert torch.allclose(lt2[:k2].t(), L2, atol=1e-9)
    assert float(lt2[k2:].abs().max()) == 0.0


def test_vector_diagonal_preconditioner_is_the_exact_inverse_of_llt_plus_d():
    K = _kmat(250)
    n = K.shape[0]
    lt = pivoted_cholesky_rows(lambda p: K[p], K.diagonal().clone(), 20, 1e-12)
    g = torch.Generator().manual_seed(3)
    d = 0.05 + 0.3 * torch.rand(n, generator=g, dtype=torch.float64)
    pre = preconditioner_from_factor(lt, n, d, torch.float64, noise_is_vector=True)
    P = lt.t() @ lt + torch.diag(d)
    R = torch.randn(7, n, generator=g, dtype=torch.float64)
    ld = pre.q1t.shape[1]
    Rt = torch.zeros(7, ld, dtype=torch.float64)
    Rt[:, :n] = R
    out = pre.apply_(Rt, torch.zeros_like(Rt))[:, :n]
    ref = torch.linalg.solve(P, R.t()).t()
    assert torch.allclose(out, ref, rtol=1e-9, atol=1e-10)
    assert abs(float(pre.logdet) - float(torch.logdet(P))) < 1e-8 * abs(float(torch.logdet(P)))
    # and the constant branch, for comparison, through the same routine
    pre_c = preconditioner_from_factor(lt, n, torch.tensor(0.2, dtype=torch.float64), torch.float64)
    Pc = lt.t() @ lt + 0.2 * torch.eye(n, dtype=torch.float64)
    out_c = pre_c.apply_(Rt, torch.zeros_like(Rt))[:, :n]
    assert torch.allclose(out_c, torch.linalg.solve(Pc, R.t()).t(), rtol=1e-9, atol=1e-10)
    assert abs(float(pre_c.logdet) - float(torch.logdet(Pc))) < 1e-8 * abs(float(torch.logdet(Pc)))
    # probe sampling covariance factor: z = L e1 + D^1/2 e2
    assert torch.allclose(pre.noise_sqrt(n, torch.float64).reshape(-1), d.sqrt())
