"""GPU parity of the split-operand product (csrc/kv_gramh.hpp: generation AND contraction on the f16 matrix pipe, both operands
split into f16 hi + lo parts) against the float64 CPU oracle, through the C ABI with GPAMD_KV_SPLIT.

The contract is the one of the fp32-MFMA kernels (tests/test_gpu_kv.py): max |d| <= 2e-5 * max |K V| per column -- the split
changes WHERE the multiply-adds run, not the stated accuracy.  Extra cases for what the split adds: columns of wildly
different magnitude (per-column power-of-two scaling), zero columns, column counts around the tile edges (5, 31..33, 64..66,
launch groups beyond 65), rectangular products, the converged-CG `done` flag, and agreement with the fp32-MFMA kernel itself.
"""
import math

import pytest
import torch

from oracle import kernels as OK

pytestmark = pytest.mark.gpu

KINDS = ["rbf", "matern32", "matern52", "rq"]


@pytest.fixture()
def split(monkeypatch):
    from gpytorch_amd import backend as B

    monkeypatch.setattr(B, "SPLIT_CONTRACTION", True)
    return B


def _oracle_K(kind, X1, X2, ls, alpha=None):
    if kind == "rbf":
        return OK.rbf(X1, X2, ls, direct=True)
    if kind == "rq":
        return (1.0 + OK.sq_dist_direct(X1 / ls, X2 / ls) / (2.0 * alpha)).pow(-alpha)
    nu = OK.KINDS[kind]
    r = (OK.sq_dist_direct(X1 / ls, X2 / ls)).sqrt() * math.sqrt(2 * nu)
    e = torch.exp(-r)
    return (1 + r) * e if nu == 1.5 else (1 + r + r * r / 3) * e


def _prep(B, kind, X, ls, dev, shift):
    return B.prep_points(kind, X.float().to(dev), torch.as_tensor(ls), shift.float().to(dev), param=1.7 if kind == "rq" else None)


def _check(B, kind, X1, X2, V, ls, dev, tol=2e-5):
    shift = torch.cat([X1, X2]).mean(0)
    p1, p2 = _prep(B, kind, X1, ls, dev, shift), _prep(B, kind, X2, ls, dev, shift)
    t, m = V.shape
    vt = torch.zeros(t, B.round_up(m, 4), device=dev)
    vt[:, :m] = V.float().to(dev)
    assert B.kv_flags(p1, p2, t) == (B.KV_GRAM | B.KV_SPLIT)
    out = B.kv(p1, p2, vt)[:, : X1.shape[0]].double().cpu()
    ref = (_oracle_K(kind, X1.float().double(), X2.float().double(), ls, 1.7) @ V.float().double().T).T
    for c in range(t):
        scale = ref[c].abs().max().clamp_min(1e-300)
        assert (out[c] - ref[c]).abs().max() <= tol * scale, (kind, c, float((out[c] - ref[c]).abs().max() / scale))
    return out


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("n,m,d,t", [(300, 300, 3, 65), (257, 513, 1, 5), (130, 1000, 10, 33), (1000, 129, 6, 32), (65, 3000, 16, 64),
                                       (700, 700, 2, 11), (129, 400, 4, 17)])
def test_split_product_vs_oracle(split, kind, n, m, d, t, dev):
    g = torch.Generator().manual_seed(n + 7 * m + t)
    X1 = torch.rand(n, d, generator=g, dtype=torch.float64)
    X2 = X1 if n == m else torch.rand(m, d, generator=g, dtype=torch.float64)
    V = torch.randn(t, m, generator=g, dtype=torch.float64)
    _check(split, kind, X1, X2, V, 0.25 + 0.12 * d, dev)


@pytest.mark.parametrize("t", [5, 31, 66, 70, 97, 129, 130, 200])
def test_split_launch_groups(split, t, dev):
    g = torch.Generator().manual_seed(t)
    X = torch.rand(777, 3, generator=g, dtype=torch.float64)
    V = torch.randn(t, 777, generator=g, dtype=torch.float64)
    _check(split, "rbf", X, X, V, 0.3, dev)


def test_split_column_scaling_extremes(split, dev):
    """Columns spanning 60 orders of magnitude, a zero column, a column with one huge outlier, a smooth positive column (no
    cancellation: the regime where a biased split would show) and a +-1 Rademacher column."""
    g = torch.Generator().manual_seed(3)
    n = 2000
    X = torch.rand(n, 3, generator=g, dtype=torch.float64)
    V = torch.randn(9, n, generator=g, dtype=torch.float64)
    V[0] *= 1e-30
    V[1] *= 1e30
    V[2] = 0.0
    V[3, 17] = 1e6
    V[4] = 1.0 + 0.1 * torch.rand(n, generator=g, dtype=torch.float64)
    V[5] = torch.sign(V[5])
    V[6] *= 1e-12
    V[7] = torch.exp(8.0 * V[7])          # log-normal: entries from 1e-14 to 1e14 in ONE column
    out = _check(split, "rbf", X, X, V, 0.2, dev)
    assert float(out[2].abs().max()) == 0.0
    # short lengthscale (max |z|^2 close to the Gram policy limit): most K entries are far below 1 -- the regime the 2^12 scale
    # of K is for
    _check(split, "rbf", X, X, V, 0.135, dev)
    _check(split, "matern52", X, X, V, 0.36, dev)


def test_split_matches_fp32_mfma_kernel(split, dev):
    """Same product on the fp32-MFMA kernel (kv_gram.hpp) and on the split path: both within 2e-5 of the oracle, and within
    1e-5 of each other (they share the squared distances; only the contraction differs)."""
    B = split
    g = torch.Generator().manual_seed(5)
    n, t = 5000, 65
    X = torch.rand(n, 3, generator=g, dtype=torch.float64)
    V = torch.randn(t, n, generator=g, dtype=torch.float64)
    p = B.prep_points("rbf", X.float().to(dev), torch.tensor(0.25), X.mean(0).float().to(dev))
    vt = torch.zeros(t, B.round_up(n, 4), device=dev)
    vt[:, :n] = V.float().to(dev)
    a = B.kv(p, p, vt).double()
    B.SPLIT_CONTRACTION = False
    b = B.kv(p, p, vt).double()
    B.SPLIT_CONTRACTION = True
    assert float((a - b).abs().max() / b.abs().max()) < 1e-5
    ref = (OK.rbf(X.float().double(), X.float().double(), 0.25, direct=True) @ V.float().double().T).T
    ea = float((a[:, :n].cpu() - ref).abs().max() / ref.abs().max())
    eb = float((b[:, :n].cpu() - ref).abs().max() / ref.abs().max())
    assert ea < 2e-5 and eb < 2e-5, (ea, eb)


def test_split_cg_solve_and_done_flag(split, dev):
    """mBCG on the split product: same solution as dense float64 Cholesky to the CG tolerance; converged solves turn the
    remaining launches (pre-pass included) into no-ops through the `done` flag."""
    from gpytorch_amd.linear_cg import linear_cg

    B = split
    g = torch.Generator().manual_seed(11)
    n, t = 1500, 9
    X = torch.rand(n, 2, generator=g, dtype=torch.float64)
    rhs = torch.randn(t, n, generator=g, dtype=torch.float64)
    p = B.prep_points("rbf", X.float().to(dev), torch.tensor(0.3), X.mean(0).float().to(dev))
    rt = torch.zeros(t, B.round_up(n, 4), device=dev)
    rt[:, :n] = rhs.float().to(dev)
    scale, noise = torch.tensor([1.3], device=dev), torch.tensor([0.05], device=dev)
    xt, info = linear_cg(p, scale, noise, rt, tolerance=1e-4, max_iter=500)
    assert info.tolerance_reached
    Kh = 1.3 * OK.rbf(X.float().double(), X.float().double(), 0.3, direct=True) + 0.05 * torch.eye(n, dtype=torch.float64)
    ref = torch.linalg.solve(Kh, rhs.float().double().T).T
    sol = xt[:, :n].double().cpu()
    assert float((sol - ref).norm() / ref.norm()) < 2e-3


@pytest.mark.parametrize("kind,d,t", [("matern32", 6, 33), ("rq", 2, 64), ("matern52", 12, 65), ("rbf", 5, 32)])
def test_split_product_large_row_tiles(split, kind, d, t, dev):
    """n >= 16 384 output rows: the two- / four-row-tile kernels (the small shapes above run one row tile per wave), checked on a
    row sample for the families and dimensions the at-size configurations do not cover."""
    B = split
    g = torch.Generator().manual_seed(d * 100 + t)
    n = 20_000
    X = torch.rand(n, d, generator=g, dtype=torch.float64)
    V = torch.randn(t, n, generator=g, dtype=torch.float64)
    ls = 0.25 + 0.12 * d
    p = _prep(B, kind, X, ls, dev, X.mean(0))
    vt = torch.zeros(t, B.round_up(n, 4), device=dev)
    vt[:, :n] = V.float().to(dev)
    assert B.kv_flags(p, p, t) == (B.KV_GRAM | B.KV_SPLIT)
    rows = torch.cat([torch.arange(0, 200), torch.randint(200, n - 200, (400,), generator=g), torch.arange(n - 200, n)])
    out = B.kv(p, p, vt)[:, rows.to(dev)].double().cpu()
    ref = (_oracle_K(kind, X.float().double()[rows], X.float().double(), ls, 1.7) @ V.float().double().T).T
    err = ((out - ref).abs().amax(1) / ref.abs().amax(1)).max()
    assert float(err) < 2e-5, (kind, float(err))


# ---- round 5: direct differences + split contraction (csrc/kv_directh.hpp; flags = GPAMD_KV_SPLIT without GPAMD_KV_GRAM) -----------------------
def _oracle_any(kind, X1, X2, ls, alpha=1.7):
    if kind == "matern12":
        return torch.exp(-OK.sq_dist_direct(X1 / ls, X2 / ls).sqrt())
    return _oracle_K(kind, X1, X2, ls, alpha)


def _check_direct(B, kind, X1, X2, V, ls, dev, tol=2e-5):
    shift = torch.cat([X1, X2]).mean(0)
    p1, p2 = _prep(B, kind, X1, ls, dev, shift), _prep(B, kind, X2, ls, dev, shift)
    t, m = V.shape
    vt = torch.zeros(t, B.round_up(m, 4), device=dev)
    vt[:, :m] = V.float().to(dev)
    out = B.kv(p1, p2, vt)[:, : X1.shape[0]].double().cpu()
    ref = (_oracle_any(kind, X1.float().double(), X2.float().double(), ls) @ V.float().double().T).T
    for c in range(t):
        scale = ref[c].abs().max().clamp_min(1e-300)
        assert (out[c] - ref[c]).abs().max() <= tol * scale, (kind, c, float((out[c] - ref[c]).abs().max() / scale))
    return out


@pytest.mark.parametrize("kind", ["rbf", "matern12", "matern32", "matern52", "rq"])
@pytest.mark.parametrize("n,m,d,t", [(300, 300, 3, 11), (257, 513, 1, 5), (130, 1000, 10, 32), (1000, 129, 6, 33), (65, 3000, 8, 64), (700, 700, 2, 65),
                                       (129, 400, 4, 17), (17_000, 600, 3, 11), (16_500, 300, 9, 8), (400, 400, 5, 40),
                                       # round 6: two column tiles + the extra VALU column (K generated once for up to 65 columns), both row tilings
                                       (16_600, 500, 3, 65), (16_400, 400, 2, 40), (16_500, 300, 8, 65), (16_400, 257, 10, 33), (300, 300, 10, 65),
                                       (200, 600, 4, 129), (500, 500, 6, 70)])
def test_direct_split_product_vs_oracle(split, kind, n, m, d, t, dev, monkeypatch):
    """Forced onto the direct-difference generation (as for a cloud outside the policy of the quadratic expansion) with the split contraction:
    groups of <= 64 (+ 1) columns on kv_directh_kernel (one or two 32-column tiles, the 33rd / 65th column on the VALU; two row tiles per wave from
    16 384 output rows on -- one with two column tiles beyond four dimensions --, one below), trailing groups of < 5 columns on the VALU kernel, every
    family incl. Matern nu = 1/2 (which has no Gram form), d up to 10."""
    monkeypatch.setattr(split, "FORCE_KV_FLAGS", split.KV_SPLIT)
    g = torch.Generator().manual_seed(n + 7 * m + t)
    X1 = torch.rand(n, d, generator=g, dtype=torch.float64)
    X2 = X1 if n == m else torch.rand(m, d, generator=g, dtype=torch.float64)
    V = torch.randn(t, m, generator=g, dtype=torch.float64)
    _check_direct(split, kind, X1, X2, V, 0.25 + 0.12 * d, dev)


def test_direct_split_is_what_a_short_lengthscale_selects(split, dev):
    """A cloud outside every Gram-form policy (a few hundred points spread over hundreds of lengthscales) and Matern nu = 1/2: ``kv_flags`` keeps the
    contraction on the f16 matrix pipe (KV_SPLIT alone), the result matches the float64 oracle and the fp32 direct kernels."""
    import warnings

    B = split
    g = torch.Generator().manual_seed(3)
    X = torch.rand(900, 3, generator=g, dtype=torch.float64)
    V = torch.randn(11, 900, generator=g, dtype=torch.float64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        p = _prep(B, "rbf", X, 0.003, dev, X.mean(0))
        assert B.kv_flags(p, p, 11) == B.KV_SPLIT and B.kv_flags(p, p, 3) == B.KV_SPLIT
        p12 = _prep(B, "matern12", X, 0.4, dev, X.mean(0))
        assert B.kv_flags(p12, p12, 11) == B.KV_SPLIT
        a = _check_direct(B, "rbf", X, X, V, 0.003, dev)
        b = _check_direct(B, "matern12", X, X, V, 0.4, dev)
        B.SPLIT_CONTRACTION = False
        try:
            assert B.kv_flags(p, p, 11) == 0
            a0 = _check_direct(B, "rbf", X, X, V, 0.003, dev)
            b0 = _check_direct(B, "matern12", X, X, V, 0.4, dev)
        finally:
            B.SPLIT_CONTRACTION = True
    assert float((a - a0).abs().max() / a0.abs().max()) < 2e-5 and float((b - b0).abs().max() / b0.abs().max()) < 2e-5
