"""GPU parity: device-resident mBCG, pivoted Cholesky, preconditioner, SLQ log-det -- HIP path
(through the C ABI) against the CPU oracle and dense float64 Cholesky.

Tolerances: BASELINE.json north_star asks rtol 1e-3 on solves / log-det (given probes) /
predictive mean+variance; the reference itself enforces rtol 0.02 on CG solves
(test/lazy/test_lazy_evaluated_kernel_tensor.py:84-105) and 1e-2 relative on log_prob
(test/distributions/test_multivariate_normal.py:219-237).
"""
import math

import pytest
import torch

from oracle import exact_gp as OG
from oracle import kernels as OK
from oracle import linear_cg as OCG
from oracle import pivoted_cholesky as OPC
from tests.util import make_data, rel_err

pytestmark = pytest.mark.gpu


def _setup(kind, n, d, ls, dev, seed=0):
    from gpytorch_amd import backend as B

    X, y = make_data(n, d, seed)
    shift = None if kind == "rbf" else X.mean(0).to(dev)
    xp = B.prep_points(kind, X.float().to(dev), torch.tensor(ls), shift)
    return X, y, xp


@pytest.mark.parametrize("kind,d,ls", [("rbf", 3, 0.25), ("matern52", 10, 0.8)])
@pytest.mark.parametrize("t", [1, 11, 65])
def test_cg_solve_vs_cholesky(kind, d, ls, t, dev):
    """Tight-tolerance CG (cg_tolerance 1e-4 as in test_lazy_evaluated_kernel_tensor.py:84) vs dense solve."""
    from gpytorch_amd import backend as B
    from gpytorch_amd.linear_cg import linear_cg

    n = 1500
    X, y, xp = _setup(kind, n, d, ls, dev)
    g = torch.Generator().manual_seed(3)
    rhs = torch.randn(n, t, generator=g, dtype=torch.float64)
    rhs[:, 0] = y
    sc = torch.tensor([1.3], device=dev)
    s2 = torch.tensor([0.1], device=dev)
    sol_t, info = linear_cg(xp, sc, s2, B.to_probe_major(rhs.to(dev)), tolerance=1e-4, max_iter=500)
    sol = B.from_probe_major(sol_t, n)
    ref, _ = OG.dense_solve_logdet(kind, X, rhs, ls, 1.3, 0.1)
    assert info.tolerance_reached
    assert rel_err(sol, ref) < 1e-3


def test_cg_iteration_parity_with_oracle(dev):
    """Same iteration count, same per-iteration alpha/beta (to fp32 rounding) as the restated
    linear_cg on identical inputs, including the tridiagonal matrices."""
    from gpytorch_amd import backend as B
    from gpytorch_amd.linear_cg import linear_cg

    kind, n, d, ls, t = "rbf", 1200, 3, 0.25, 9
    X, y, xp = _setup(kind, n, d, ls, dev)
    g = torch.Generator().manual_seed(11)
    rhs = torch.randn(n, t, generator=g, dtype=torch.float64)
    rhs[:, -1] = 0.0  # a zero right-hand side column must come back as exact zeros
    sc = torch.tensor([1.0], device=dev)
    s2 = torch.tensor([0.1], device=dev)
    sol_t, info = linear_cg(xp, sc, s2, B.to_probe_major(rhs.to(dev)), n_tridiag=t - 2, tolerance=1e-3, max_iter=200)
    # the reference runs CG in the dtype of its inputs: restate in float32 for iteration-level parity
    mm = OG.make_matmul(kind, X.float(), ls, 1.0, 0.1)
    ref, Tref, oinfo = OCG.linear_cg(mm, rhs.float(), n_tridiag=t - 2, tolerance=1e-3, max_iter=200, return_info=True)
    # the stopping iteration of a float32 CG at tolerance 1e-3 moves by a few steps with the summation order
    # of K*V (direct vs Gram-form generation, CPU BLAS vs MFMA): same algorithm, +-3 % on ~90 iterations
    assert abs(info.iterations - oinfo["iters"]) <= 3
    assert info.tolerance_reached == oinfo["tolerance_reached"]
    sol = B.from_probe_major(sol_t, n)
    assert torch.equal(sol[:, -1].cpu(), torch.zeros(n))
    assert rel_err(sol, ref) < 5e-3
    assert info.t_mats.shape == Tref.shape
    # Lanczos/CG coefficients are chaotic in finite precision: late rows of T differ between any two
    # float32 implementations (summation order), the early rows and the quadrature they feed do not
    assert rel_err(info.t_mats[:, :8, :8], Tref[:, :8, :8]) < 5e-3
    from oracle import slq as OS
    from gpytorch_amd.bbmm import slq_logdet
    assert abs(float(slq_logdet(info.t_mats, n)) - float(OS.slq_logdet(Tref, n))) < 2e-3 * abs(float(OS.slq_logdet(Tref, n)))
    # and the float64 restatement agrees on the solution to the CG tolerance
    ref64 = OCG.linear_cg(OG.make_matmul(kind, X, ls, 1.0, 0.1), rhs, tolerance=1e-3, max_iter=200)
    assert rel_err(sol, ref64) < 5e-3


def test_cg_default_tolerance_stops_like_reference(dev):
    """cg_tolerance = 1 (training default): >= 10 iterations (>= 20 with tridiagonals), then the first
    iteration whose mean relative residual is < 1 -- same iteration as the float32 restatement."""
    from gpytorch_amd import backend as B
    from gpytorch_amd.linear_cg import linear_cg

    kind, n, d, ls = "rbf", 1000, 3, 0.25
    X, y, xp = _setup(kind, n, d, ls, dev)
    rhs = torch.randn(n, 5, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    sc, s2 = torch.tensor([1.0], device=dev), torch.tensor([0.1], device=dev)
    mm = OG.make_matmul(kind, X.float(), ls, 1.0, 0.1)
    _, info = linear_cg(xp, sc, s2, B.to_probe_major(rhs.to(dev)), n_tridiag=4, tolerance=1.0)
    _, _, oi = OCG.linear_cg(mm, rhs.float(), n_tridiag=4, tolerance=1.0, return_info=True)
    assert info.iterations == oi["iters"] >= 21 and info.t_mats.shape[-1] == 20
    _, info = linear_cg(xp, sc, s2, B.to_probe_major(rhs.to(dev)), n_tridiag=0, tolerance=1.0)
    _, oi = OCG.linear_cg(mm, rhs.float(), n_tridiag=0, tolerance=1.0, return_info=True)
    assert info.iterations == oi["iters"] >= 11


@pytest.mark.parametrize("kind,d,ls", [("rbf", 3, 0.25), ("matern52", 10, 0.8), ("matern32", 2, 0.3)])
def test_pivoted_cholesky_vs_oracle(kind, d, ls, dev):
    from gpytorch_amd import backend as B

    n, rank = 900, 40
    X, y, xp = _setup(kind, n, d, ls, dev)
    sc = torch.tensor([1.5], device=dev)
    Lt, piv, m = B.pivoted_cholesky(xp, sc, rank, 1e-3)
    diag = torch.full((n,), 1.5, dtype=torch.float64)
    Kd = OK.kernel_matrix(kind, X, X, ls, 1.5, x1_eq_x2=True, direct=True)
    # float32 makes exact ties in the running diagonal common (every point far from all pivots keeps
    # d_i == theta), so the pivot SEQUENCE is only defined up to rounding: replay the device's pivots
    # in the float64 oracle and require (a) each was an arg-max up to fp32 rounding, (b) same L.
    Lref, _, gaps = OPC.pivoted_cholesky(diag, lambda p: Kd[p], rank, 1e-3, forced_pivots=piv.cpu(), return_gaps=True)
    assert m == Lref.shape[1] == rank
    assert len(set(piv.cpu().tolist())) == m
    assert max(gaps) < 1e-5, gaps
    assert rel_err(Lt.t(), Lref) < 1e-3
    # identical tie-breaking to the sequential reference when the arithmetic is identical: float32 oracle
    Kf = Kd.float()
    L32, p32 = OPC.pivoted_cholesky(diag.float(), lambda p: Kf[p], rank, 1e-3, return_pivots=True)
    assert int(piv[0]) == int(p32[0]) == 0  # step 0: all-equal diagonal -> first position wins
    # property: residual trace shrinks monotonically and K - L L^T stays PSD-ish
    res = Kd - (Lt.t().double().cpu() @ Lt.double().cpu())
    assert res.diagonal().min() > -1e-4


def test_pivoted_cholesky_early_stop(dev):
    """A long lengthscale makes K numerically low-rank: the error test must stop before `rank`."""
    from gpytorch_amd import backend as B

    n = 600
    X, y, xp = _setup("rbf", n, 2, 3.0, dev)
    Lt, piv, m = B.pivoted_cholesky(xp, None, 60, 1e-3)
    Kd = OK.kernel_matrix("rbf", X, X, 3.0, 1.0, x1_eq_x2=True, direct=True)
    Lref = OPC.pivoted_cholesky(torch.ones(n, dtype=torch.float64), lambda p: Kd[p], 60, 1e-3)
    assert abs(m - Lref.shape[1]) <= 1 and m < 60


def test_preconditioner_matches_dense_inverse(dev):
    from gpytorch_amd import backend as B
    from gpytorch_amd.bbmm import build_preconditioner

    n = 2500
    X, y, xp = _setup("rbf", n, 3, 0.25, dev)
    sc, s2 = torch.tensor([1.0], device=dev), torch.tensor([0.1], device=dev)
    pre = build_preconditioner(xp, sc, s2, rank=30, tol=1e-3, min_size=2000)
    assert pre is not None
    L = pre.lt[:, :n].t().double().cpu()
    P = L @ L.t() + 0.1 * torch.eye(n, dtype=torch.float64)
    V = torch.randn(n, 7, generator=torch.Generator().manual_seed(2), dtype=torch.float64)
    vt = B.to_probe_major(V.to(dev))
    out = torch.zeros_like(vt)
    pre.apply_(vt, out)
    ref = torch.linalg.solve(P, V)
    assert rel_err(B.from_probe_major(out, n), ref) < 1e-3
    assert abs(float(pre.logdet) - float(torch.linalg.slogdet(P)[1])) < 1e-3 * abs(float(torch.linalg.slogdet(P)[1]))
    assert build_preconditioner(xp, sc, s2, rank=0) is None
    assert build_preconditioner(xp, sc, s2, rank=15, min_size=5000) is None
    # the fused second half (gpamd_precond_apply_f32f64) against the same formula in torch float64, several column groups
    R = torch.randn(37, vt.shape[1], generator=torch.Generator().manual_seed(3)).to(dev)
    R[:, n:] = 0
    got = pre.apply_(R, torch.empty_like(R)).double()
    q = pre.q1t.double()
    want = (R.double() - (R.double() @ q.t()) @ q) / 0.1
    assert float((got - want).abs().max() / want.abs().max()) < 1e-6


def _probes(n, t, L, s2, seed=1234):
    """Probe matrix shared by both sides: N(0, I) without a preconditioner, N(0, L L^T + s2 I) with one
    (SURVEY.md A.5 -- the SLQ / trace estimators are only unbiased for probes whose covariance is the
    preconditioner actually applied, so L is the DEVICE's pivoted-Cholesky factor)."""
    g = torch.Generator().manual_seed(seed)
    if L is None:
        return torch.randn(n, t, generator=g, dtype=torch.float64)
    return L @ torch.randn(L.shape[1], t, generator=g, dtype=torch.float64) + math.sqrt(s2) * torch.randn(n, t, generator=g, dtype=torch.float64)


@pytest.mark.parametrize("precond_rank", [0, 20])
def test_inv_quad_logdet_given_probes(precond_rank, dev):
    """Same probe matrix Z on both sides: inv_quad and SLQ log-det match the oracle to 1e-3 and
    dense Cholesky to the stochastic accuracy of the estimator."""
    from gpytorch_amd import backend as B
    from gpytorch_amd import settings
    from gpytorch_amd.bbmm import build_preconditioner, inv_quad_logdet_forward

    kind, n, d, ls, t = "rbf", 2200, 3, 0.25, 32
    X, y, xp = _setup(kind, n, d, ls, dev)
    sc, s2 = torch.tensor([1.0], device=dev), torch.tensor([0.1], device=dev)
    pre = build_preconditioner(xp, sc, s2, rank=precond_rank, tol=1e-3, min_size=2000)
    Ldev = None if pre is None else pre.lt[:, :n].t().double().cpu()
    Z = _probes(n, t, Ldev, 0.1)
    res = inv_quad_logdet_forward(xp, sc, s2, B.to_probe_major(y.unsqueeze(-1).to(dev)), precond=pre, probes=Z, tolerance=1e-4)
    # oracle with identical probes (and, when preconditioned, the identical L via its own pivoted Cholesky)
    # float32 restatement (what the reference executes for float32 inputs): rtol 1e-3; float64: 3e-3
    for dt, tol in ((torch.float32, 1e-3), (torch.float64, 3e-3)):
        mll, aux = OG.bbmm_mll(kind, X.to(dt), y.to(dt), ls, 1.0, 0.1, precond_rank=precond_rank, min_precond_size=2000,
                               cg_tol=1e-4, probes=Z.to(dt), return_aux=True, precond_L=Ldev)
        assert abs(float(res.inv_quad.sum()) - float(aux["inv_quad"])) < tol * abs(float(aux["inv_quad"]))
        assert abs(float(res.logdet) - float(aux["logdet"])) < tol * abs(float(aux["logdet"]))
    _, ld_exact = OG.dense_solve_logdet(kind, X, y.unsqueeze(-1), ls, 1.0, 0.1)
    assert abs(float(res.logdet) - float(ld_exact)) < 0.05 * abs(float(ld_exact))


def test_lanczos_properties_and_parity(dev):
    """Device Lanczos (full re-orthogonalisation): Q orthonormal, Q K_hat Q^T = T, early T block equal to the
    float64 restatement started from the same vector."""
    from gpytorch_amd import backend as B
    from gpytorch_amd.lanczos import lanczos_tridiag
    from oracle import lanczos as OL

    kind, n, d, ls, k = "matern52", 1200, 4, 0.5, 60
    X, y, xp = _setup(kind, n, d, ls, dev)
    sc, s2 = torch.tensor([1.2], device=dev), torch.tensor([0.1], device=dev)
    init = torch.randn(n, 1, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    Qt, T = lanczos_tridiag(xp, sc, s2, k, B.to_probe_major(init.to(dev)))
    Q = Qt[:, :n].double().cpu()
    assert Q.shape[0] == k
    assert (Q @ Q.t() - torch.eye(k, dtype=torch.float64)).abs().max() < 1e-4
    Kh = OK.kernel_matrix(kind, X, X, ls, 1.2, x1_eq_x2=True, direct=True) + 0.1 * torch.eye(n, dtype=torch.float64)
    assert rel_err(Q @ Kh @ Q.t(), T) < 1e-3
    Qr, Tr = OL.lanczos_tridiag(lambda v: Kh @ v, k, n, init)
    assert rel_err(T[:6, :6], Tr[:6, :6]) < 1e-3


def test_full_size_solve_residual_symmetry_and_rows(dev):
    """Size-independent properties at BASELINE sizes (the oracle cannot hold these): configs[1] n = 100 000 --
    (i) the operator is symmetric, u^T (K v) == v^T (K u); (ii) the mBCG solution really solves the system: the TRUE
    relative residual |K_hat x - b| / |b|, recomputed with one more fused product, is at the requested tolerance;
    (iii) metric size n = 500 000 -- K e_j reproduces explicitly generated rows of K."""
    from gpytorch_amd import backend as B
    from gpytorch_amd.linear_cg import linear_cg

    n, d, t = 100_000, 3, 4
    g = torch.Generator().manual_seed(0)
    X = torch.rand(n, d, generator=g)
    y = torch.sin(2 * math.pi * X[:, 0]) + torch.cos(math.pi * X.sum(-1)) + 0.1 * torch.randn(n, generator=g)
    xp = B.prep_points("rbf", X.to(dev), torch.tensor(0.25), X.mean(0).to(dev))
    sc, s2 = torch.tensor([1.0], device=dev), torch.tensor([0.1], device=dev)
    ld = B.round_up(n, 4)
    U = torch.zeros(t, ld, device=dev)
    V = torch.zeros(t, ld, device=dev)
    U[:, :n] = torch.randn(t, n, generator=g).to(dev)
    V[:, :n] = torch.randn(t, n, generator=g).to(dev)
    ku, kv_ = B.kv(xp, xp, U).clone(), B.kv(xp, xp, V).clone()
    a, b = (V.double() * ku.double()).sum(-1), (U.double() * kv_.double()).sum(-1)
    scale = (V.double().abs() * ku.double().abs()).sum(-1)  # the sums cancel heavily: errors scale with the absolute terms
    assert float(((a - b).abs() / scale).max()) < 2e-5
    rhs = torch.zeros(t, ld, device=dev)
    rhs[:, :n] = torch.cat([y.unsqueeze(0), torch.randn(t - 1, n, generator=g)], 0).to(dev)
    with torch.no_grad():
        sol, info = linear_cg(xp, sc, s2, rhs, tolerance=1e-3, max_iter=1000)
    assert info.tolerance_reached
    res = B.kv(xp, xp, sol, scale=sc, dscale=s2, vd=sol)[:, :n] - rhs[:, :n]
    rel = res.norm(dim=-1) / rhs[:, :n].norm(dim=-1)
    assert float(rel.mean()) < 1.5e-3, rel.tolist()
    # (iii) n = 500 000
    n5 = 500_000
    X5 = torch.rand(n5, d, generator=g)
    p5 = B.prep_points("rbf", X5.to(dev), torch.tensor(0.25), X5.mean(0).to(dev))
    idx = torch.tensor([0, 77_777, 250_001, n5 - 1])
    E = torch.zeros(len(idx), B.round_up(n5, 4), device=dev)
    E[torch.arange(len(idx)), idx] = 1.0
    cols = B.kv(p5, p5, E)[:, :n5]
    rows = B.kernel_rows(p5, idx, p5)
    assert float((cols - rows).abs().max()) < 5e-6


@pytest.mark.parametrize("precond_rank", [0, 10])
def test_cg_graph_replay_is_bitwise_the_eager_loop(precond_rank, dev):
    """Launch-bound solves record one mBCG iteration into a hipGraph and replay it (settings.cg_graph): same kernels, same order,
    the iteration index read from the device -- solution, iteration count and the recorded alpha / beta tridiagonals are
    bitwise those of the eager loop, with and without the pivoted-Cholesky preconditioner."""
    from gpytorch_amd import backend as B
    from gpytorch_amd import settings
    from gpytorch_amd.bbmm import build_preconditioner
    from gpytorch_amd.linear_cg import linear_cg

    g = torch.Generator().manual_seed(21)
    n, t = 3000, 11
    X = torch.rand(n, 2, generator=g)
    rhs = torch.randn(t, n, generator=g)
    xp = B.prep_points("rbf", X.to(dev), torch.tensor(0.2), X.mean(0).to(dev))
    rt = torch.zeros(t, B.round_up(n, 4), device=dev)
    rt[:, :n] = rhs.to(dev)
    scale, noise = torch.tensor([1.0], device=dev), torch.tensor([0.01], device=dev)
    pc = build_preconditioner(xp, scale, noise, rank=precond_rank, min_size=0) if precond_rank else None
    out = {}
    for graph in (False, True):
        with settings.cg_graph(graph):
            xt, info = linear_cg(xp, scale, noise, rt, n_tridiag=t - 1, tolerance=1e-3, max_iter=400, preconditioner=pc)
        out[graph] = (xt.clone(), info)
    (xa, ia), (xb, ib) = out[False], out[True]
    assert ia.iterations == ib.iterations and ia.tolerance_reached and ib.tolerance_reached and ia.iterations > 12
    assert torch.equal(xa, xb)
    assert torch.equal(ia.t_mats, ib.t_mats)
