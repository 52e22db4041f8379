"""GPU: the BLOCK-CENTRED Gram expansion -- the fast kernels beyond max |x / lengthscale|^2 <= 32.

Round 2 kept the Gram-form generation and the split-operand contraction inside the accuracy bound of a CLOUD-centred quadratic expansion
(max |z|^2 <= 32) and dropped to the direct-difference kernel (2.4-3x slower) outside it: short lengthscales on U[0,1]^d, standardised
inputs at l < 0.8 sqrt(d) -- most real training runs.  The reference's Gram-trick distance has no scale limit
(``gpytorch/kernels/kernel.py:26-49``; KeOps twin ``kernels/keops/rbf_kernel.py:12-15``).  Now the rows of x1 are sorted along a Hilbert curve
and every workgroup expands the squared distances around the centre of its own row block (csrc/gram_f16.hpp ``load_center``;
``backend.gram_mode`` = 2), so the cancellation error scales with the block radius.  Ground truth: the float64 oracle
(reference dense formulas) and dense float64 algebra.
"""
import math
import warnings

import pytest
import torch

from oracle import exact_gp as OG
from oracle import kernels as OK
from tests.util import make_data, rel_err

pytestmark = pytest.mark.gpu


def _cloud(name, n, d=3, seed=0):
    g = torch.Generator().manual_seed(seed)
    if name == "uniform":
        return torch.rand(n, d, generator=g, dtype=torch.float32)
    return torch.randn(n, d, generator=g, dtype=torch.float32)       # "standardised" inputs


CASES = {
    # name: cloud, kind, lengthscale        (n = 60 000, d = 3: max |z|^2 ~ 200 -- far outside the cloud-centred bound of 32)
    "uniform_rbf_l0.05": ("uniform", "rbf", 0.05),            # every 512-point run of the Hilbert order is compact
    "normal_rbf_l0.3": ("normal", "rbf", 0.3),                # Gaussian tails: ~12 % of the rows on the direct-difference kernel
    "normal_matern52_l0.8": ("normal", "matern52", 0.8),
    "uniform_matern32_l0.08": ("uniform", "matern32", 0.08),
}
N_CASE = 60_000
_KROWS = {}


def _rows_and_kernel(case):
    """Sampled rows (first / last blocks + random) of the float64 reference kernel matrix, once per case."""
    if case not in _KROWS:
        cloud, kind, ls = CASES[case]
        X = _cloud(cloud, N_CASE)
        g = torch.Generator().manual_seed(7)
        rows = torch.cat([torch.arange(200), torch.randint(200, N_CASE - 200, (300,), generator=g), torch.arange(N_CASE - 200, N_CASE)]).unique()
        _KROWS[case] = (X, rows, OK.kernel_matrix(kind, X[rows].double(), X.double(), ls, 1.0, x1_eq_x2=False, direct=True))
    return _KROWS[case]


@pytest.mark.parametrize("split", [False, True], ids=["f32mfma", "split"])
@pytest.mark.parametrize("case", list(CASES))
def test_wide_clouds_stay_on_the_gram_kernels(case, split, dev, monkeypatch):
    from gpytorch_amd import backend as B

    monkeypatch.setattr(B, "SPLIT_CONTRACTION", split)
    cloud, kind, ls = CASES[case]
    n = N_CASE
    X, rows, Krows = _rows_and_kernel(case)
    Xd = X.to(dev)
    xp = B.prep_points(kind, Xd, torch.tensor([ls]), Xd.mean(0))
    assert xp.zmax2 > 32.0                                    # outside the cloud-centred bound ...
    assert B.gram_mode(xp, xp) == 2                           # ... inside the block-centred one
    sv = xp.sorted_view()
    assert sv.r2 <= B.GRAM_MAX_BLOCK_SQRADIUS and (cloud == "uniform" or sv.n_compact < n)   # Gaussian tails: wide rows on the direct kernel
    for t in (1, 2, 4, 8, 11, 16, 17, 33, 65):
        assert B.kv_flags(xp, xp, t) == (B.KV_GRAM | (B.KV_SPLIT if split else 0))
        V = torch.randn(n, t, generator=torch.Generator().manual_seed(t))
        out_t = B.kv(xp, xp, B.to_probe_major(V.to(dev)))
        got = out_t[:, rows.to(dev)].t().double().cpu()
        ref = Krows @ V.double()
        assert rel_err(got, ref) < 2e-5, (case, t, rel_err(got, ref))
    # the K entries themselves (dense rows through the SAME fused product: V = unit vectors of a few columns j near the sampled rows)
    cols = rows[:64]
    E = torch.zeros(n, cols.numel())
    E[cols, torch.arange(cols.numel())] = 1.0
    got = B.kv(xp, xp, B.to_probe_major(E.to(dev)))[:, rows.to(dev)].t().double().cpu()
    assert float((got - Krows[:, cols]).abs().max()) < 2e-5


def test_rq_wide_cloud_and_rectangular_product(dev):
    from gpytorch_amd import backend as B

    n, m, ls, alpha = 9000, 5000, 0.06, 1.3
    X1, X2 = _cloud("uniform", n, seed=1), _cloud("uniform", m, seed=2)
    sh = X1.mean(0).to(dev)
    p1 = B.prep_points("rq", X1.to(dev), torch.tensor([ls]), sh, alpha)
    p2 = B.prep_points("rq", X2.to(dev), torch.tensor([ls]), sh, alpha)
    assert B.gram_mode(p1, p2) == 2
    for t in (3, 12, 40):
        V = torch.randn(m, t, generator=torch.Generator().manual_seed(t))
        got = B.from_probe_major(B.kv(p1, p2, B.to_probe_major(V.to(dev))), n)
        ref = OK.rq(X1.double(), X2.double(), ls, alpha, x1_eq_x2=False, direct=True) @ V.double()
        assert rel_err(got, ref) < 3e-5, t
    # epilogue (scale, noise, per-point diagonal) after the rows were taken back to the original order
    V = torch.randn(n, 6, generator=torch.Generator().manual_seed(9))
    dvec = torch.rand(B.round_up(n, 4), generator=torch.Generator().manual_seed(3)).to(dev)
    vt = B.to_probe_major(V.to(dev))
    got = B.from_probe_major(B.kv(p1, p1, vt, scale=torch.tensor([1.7], device=dev), dscale=torch.tensor([0.3], device=dev), vd=vt, dvec=dvec), n)
    Kd = OK.rq(X1.double(), X1.double(), ls, alpha, x1_eq_x2=True, direct=True)
    ref = 1.7 * (Kd @ V.double()) + (0.3 + dvec[:n].double().cpu()).unsqueeze(-1) * V.double()
    assert rel_err(got, ref) < 3e-5


def test_cg_solve_and_mll_on_a_wide_cloud(dev):
    """mBCG with the Hilbert-ordered slabs (linear_cg takes the rows back every iteration) and the whole MLL through the model API, with
    gradients incl. the inputs (kv_grad2 on the sorted rows), against dense float64."""
    import gpytorch_amd as g
    from gpytorch_amd import backend as B

    n, d, ls = 4096, 2, 0.06
    X, y = make_data(n, d)
    Xd = X.float().to(dev).requires_grad_(True)

    class M(g.models.ExactGP):
        def __init__(self, x, yy, lik):
            super().__init__(x, yy, lik)
            self.mean_module = g.means.ZeroMean()
            self.covar_module = g.kernels.ScaleKernel(g.kernels.RBFKernel())

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    lik = g.likelihoods.GaussianLikelihood().to(dev)
    m = M(Xd, y.float().to(dev), lik).to(dev)
    m.covar_module.base_kernel.lengthscale, m.covar_module.outputscale, lik.noise = ls, 1.2, 0.05
    op = lik(m.train()(Xd)).lazy_covariance_matrix
    p1, _ = op.kernel_op.prepared()
    assert B.gram_mode(p1, p1) == 2
    Kh = 1.2 * OK.rbf(X, X, ls, x1_eq_x2=True, direct=True) + 0.05 * torch.eye(n, dtype=torch.float64)
    S = g.settings
    with torch.no_grad(), S.max_cholesky_size(0), S.cg_tolerance(1e-4), S.max_preconditioner_size(0):
        sol = op.solve(y.float().to(dev).unsqueeze(-1))
    assert rel_err(sol, torch.linalg.solve(Kh, y.unsqueeze(-1))) < 1e-3
    mll = g.ExactMarginalLogLikelihood(lik, m)
    lik.train()
    with S.max_cholesky_size(0), S.cg_tolerance(1e-5), S.num_trace_samples(300), S.max_preconditioner_size(0), S.max_lanczos_quadrature_iterations(60):
        torch.manual_seed(0)
        val = mll(m(Xd), m.train_targets)
        val.backward()
    p = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in (ls, 1.2, 0.05)]
    ref = OG.dense_log_prob(p[1] * OK.rbf(X, X, p[0], x1_eq_x2=True, direct=True) + p[2] * torch.eye(n, dtype=torch.float64), y) / n
    gref = torch.autograd.grad(ref, p)
    assert abs(float(val) - float(ref)) < 1e-2 * max(1.0, abs(float(ref)))   # 300-probe trace estimate
    sp = lambda v: 1.0 - math.exp(-v)  # noqa: E731
    got = torch.tensor([float(m.covar_module.base_kernel.raw_lengthscale.grad.sum()), float(m.covar_module.raw_outputscale.grad.sum()),
                        float(lik.noise_covar.raw_noise.grad.sum())], dtype=torch.float64)
    want = torch.tensor([float(gref[0]) * sp(ls), float(gref[1]) * sp(1.2), float(gref[2]) * sp(0.05 - 1e-4)], dtype=torch.float64)
    assert float((got - want).norm() / want.norm()) < 0.1, (got, want)
    # input gradients: the inverse quadratic form alone is deterministic on the BBMM branch (tests/test_gpu_grad2.py) -- two fused
    # derivative passes on the sorted rows, results taken back to the original order
    Xd.grad = None
    with S.max_cholesky_size(0), S.cg_tolerance(1e-5), S.max_preconditioner_size(0), S.debug(False):
        iq, _ = lik(m(Xd)).lazy_covariance_matrix.inv_quad_logdet(m.train_targets.unsqueeze(-1), logdet=False)
        (gx,) = torch.autograd.grad(iq, [Xd])
    X64 = X.clone().requires_grad_(True)
    Kh64 = 1.2 * OK.rbf(X64, X64, ls, x1_eq_x2=False, direct=True) + 0.05 * torch.eye(n, dtype=torch.float64)
    iq_ref = (y * torch.linalg.solve(Kh64, y)).sum()
    (gxr,) = torch.autograd.grad(iq_ref, [X64])
    assert abs(float(iq) - float(iq_ref)) < 1e-3 * abs(float(iq_ref))
    assert float((gx.double().cpu() - gxr).norm() / gxr.norm()) < 3e-3


def test_posterior_with_love_on_a_wide_cloud(dev):
    """Mean-cache CG + LOVE Lanczos fused into two-column products (operators.solve_and_root_inv) with block-centred slabs."""
    import gpytorch_amd as g

    n, ns, ls = 4096, 200, 0.06
    X, y = make_data(n + ns, 2)
    Xt, yt, Xs = X[:n], y[:n], X[n:]

    class M(g.models.ExactGP):
        def __init__(self, x, yy, lik):
            super().__init__(x, yy, lik)
            self.mean_module = g.means.ZeroMean()
            self.covar_module = g.kernels.ScaleKernel(g.kernels.RBFKernel())

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    lik = g.likelihoods.GaussianLikelihood().to(dev)
    m = M(Xt.float().to(dev), yt.float().to(dev), lik).to(dev)
    m.covar_module.base_kernel.lengthscale, m.covar_module.outputscale, lik.noise = ls, 1.0, 0.1
    m.eval()
    lik.eval()
    S = g.settings
    torch.manual_seed(1)
    # (a short lengthscale leaves K_hat with a flat spectrum: LOVE needs a large Lanczos rank here, as it does in the reference)
    with torch.no_grad(), S.max_cholesky_size(0), S.eval_cg_tolerance(1e-4), S.fast_pred_var(), S.max_root_decomposition_size(1500), S.max_preconditioner_size(0):
        pred = m(Xs.float().to(dev))
        mu, var = pred.mean, pred.variance
    Kh = OK.rbf(Xt, Xt, ls, x1_eq_x2=True, direct=True) + 0.1 * torch.eye(n, dtype=torch.float64)
    Ks = OK.rbf(Xs, Xt, ls, x1_eq_x2=False, direct=True)
    Lc = torch.linalg.cholesky(Kh)
    mu_ref = Ks @ torch.cholesky_solve(yt.unsqueeze(-1), Lc).squeeze(-1)
    var_ref = 1.0 - torch.linalg.solve_triangular(Lc, Ks.t(), upper=False).pow(2).sum(0)
    assert rel_err(mu, mu_ref) < 2e-3
    assert float((var.double().cpu() - var_ref).abs().max()) < 0.02        # prior variance 1
    m.train()
    m.eval()
    with torch.no_grad(), S.max_cholesky_size(0), S.eval_cg_tolerance(1e-5), S.max_preconditioner_size(0):   # exact variances: 200-column solves
        var_x = m(Xs.float().to(dev)).variance
    assert float((var_x.double().cpu() - var_ref).abs().max()) < 2e-3


def test_fallback_outside_the_block_bound_warns(dev):
    """A handful of points spread over hundreds of lengthscales: no run of the Hilbert order is compact -> direct-difference kernels, with a warning."""
    from gpytorch_amd import backend as B

    B._warned_fallback.clear()
    X = _cloud("uniform", 600)
    xp = B.prep_points("rbf", X.to(dev), torch.tensor([0.002]), X.mean(0).to(dev))
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        assert B.kv_flags(xp, xp, 11) & B.KV_GRAM == 0   # (the contraction may still take the f16 matrix pipe: KV_SPLIT alone = direct differences + split contraction)
    assert any("direct-difference" in str(w.message) for w in rec)
    V = torch.randn(600, 3, generator=torch.Generator().manual_seed(0))
    got = B.from_probe_major(B.kv(xp, xp, B.to_probe_major(V.to(dev))), 600)
    ref = OK.rbf(X.double(), X.double(), 0.002, x1_eq_x2=True, direct=True) @ V.double()
    assert rel_err(got, ref) < 2e-5


@pytest.mark.parametrize("split", [False, True], ids=["f32W", "splitW"])
def test_backward_on_a_heavy_tailed_cloud_sends_the_wide_rows_to_the_direct_path(split, dev):
    """Round-3 advisor finding: ``backend.kv_grad2`` ran the Gram-form derivative kernel over ALL sorted rows, including the wide groups
    (sparse tails whose block radius is outside the 2e-5 policy of the expansion) that the FORWARD already sends to the direct-difference kernel.
    Now the backward splits the same way (compact rows: kv_grad2; wide rows: the direct row-block path); hyper-parameter sums and input
    gradients on a Student-t-like cloud against float64 autograd (the reference's dense formulas)."""
    from gpytorch_amd import backend as B

    n, d, t, ls = 12288, 2, 30, 0.5
    g = torch.Generator().manual_seed(5)
    X = torch.randn(n, d, generator=g, dtype=torch.float64).clamp_(-3.5, 3.5)
    X = X * (1.0 + 3.0 * (torch.rand(n, 1, generator=g, dtype=torch.float64) < 0.08).double() * torch.rand(n, 1, generator=g, dtype=torch.float64))   # 8 % outliers, up to 4 x
    Lv = torch.randn(n, t, generator=g, dtype=torch.float64).abs()
    Rv = torch.randn(n, t, generator=g, dtype=torch.float64).abs()
    xp = B.prep_points("rbf", X.float().to(dev), torch.tensor([ls]), X.float().to(dev).mean(0))
    assert B.gram_mode(xp, xp) == 2
    sv = xp.sorted_view()
    assert 0 < xp.n - sv.n_compact < 0.25 * n          # there ARE wide rows
    B.SPLIT_CONTRACTION = split
    try:
        out, gz = B.kv_grad2(xp, xp, B.to_probe_major(Lv.float().to(dev)), B.to_probe_major(Rv.float().to(dev)), iso=False, want_gz1=True)
        out0, _ = B.kv_grad2(xp, xp, B.to_probe_major(Lv.float().to(dev)), B.to_probe_major(Rv.float().to(dev)), iso=True)
    finally:
        B.SPLIT_CONTRACTION = None
    # float64 truth (on the device) on the PREPARED coordinates z: K = 2^-|zi - zj|^2, sum W K, per-dimension sums and d/dz_i
    z = xp.xp[:, :d].double().requires_grad_(True)
    z2 = z.detach().clone()
    S = (z.unsqueeze(1) - z2.unsqueeze(0)).pow(2)
    K = torch.exp2(-S.sum(-1))
    W = Lv.to(dev) @ Rv.to(dev).t()
    tot = (W * K).sum()
    (gzr,) = torch.autograd.grad(tot, [z])
    gzr = gzr.cpu()
    A = (W * K * (-math.log(2.0))).detach()
    gq = (A.unsqueeze(-1) * S.detach()).sum((0, 1)).cpu()
    tot = tot.detach().cpu()
    del S, K, W, A
    assert abs(float(out[0]) - float(tot)) < 2e-5 * abs(float(tot))
    assert float((out[1 : 1 + d].double().cpu() - gq).abs().max() / gq.abs().max()) < 2e-5
    assert abs(float(out0[1]) - float(gq.sum())) < 2e-5 * abs(float(gq.sum()))
    # every row, the tail rows included (they were the ones outside the accuracy policy)
    err = (gz.double().cpu() - gzr).norm(dim=1) / gzr.norm(dim=1).clamp_min(1e-12 * float(gzr.norm(dim=1).max()))
    tail = sv.perm[sv.n_compact :].cpu()
    assert float(err[tail].max()) < 1e-4, float(err[tail].max())
    assert float((gz.double().cpu() - gzr).norm() / gzr.norm()) < 2e-5


def _road_cloud(n, seed=3):
    """Points along a few smooth planar curves with a slowly varying third coordinate (locally one-dimensional, like a road network)."""
    g = torch.Generator().manual_seed(seed)
    roads = 12
    per = (n + roads - 1) // roads
    t = torch.linspace(0, 1, per, dtype=torch.float64).unsqueeze(0)
    p0 = torch.rand(roads, 1, 2, generator=g, dtype=torch.float64) * 3.0
    ang = torch.rand(roads, 1, generator=g, dtype=torch.float64) * 2 * math.pi
    curv = (torch.rand(roads, 1, generator=g, dtype=torch.float64) - 0.5) * 5.0
    th = ang + curv * t
    step = 1.5 / per
    xy = p0 + torch.stack([torch.cumsum(torch.cos(th) * step, 1), torch.cumsum(torch.sin(th) * step, 1)], -1)
    xy = xy.reshape(-1, 2)[:n]
    X = torch.cat([xy, (torch.sin(0.7 * xy[:, 0]) * torch.cos(0.5 * xy[:, 1])).unsqueeze(-1)], -1)
    X = X + 1e-3 * torch.randn(X.shape, generator=g, dtype=torch.float64)
    return X[torch.randperm(n, generator=g)].float().contiguous()


@pytest.mark.parametrize("split", [False, True], ids=["f32mfma", "split"])
@pytest.mark.parametrize("kind,ls", [("matern52", 0.04), ("rbf", 0.015)])
def test_elongated_runs_take_the_128_row_blocks(kind, ls, split, dev, monkeypatch):
    """Round 5.  Points along curves at a short lengthscale: 512-row runs of the Hilbert order are elongated (radius outside the policy), their
    128-row chunks are not -- ``SortedView``'s MEDIUM region, served by the kernels that centre 128-row blocks (the split kernel at one row tile per
    wave, flag GPAMD_KV_BLOCK128; every other column group falls to the direct-difference kernels inside the library).  The cloud's extent
    (max |z|^2 ~ 24 000) is also beyond the f16 range of the split norms: they saturate (csrc/gram_f16.hpp gram_norm_clamp).  Every column-count
    kernel against float64 rows, plus the backward (the derivative kernel centres 128-row blocks)."""
    from gpytorch_amd import backend as B

    monkeypatch.setattr(B, "SPLIT_CONTRACTION", split)
    n = 30_000
    X = _road_cloud(n)
    Xd = X.to(dev)
    xp = B.prep_points(kind, Xd, torch.tensor([ls]), Xd.mean(0))
    with warnings.catch_warnings():
        warnings.simplefilter("error")                      # no fall-back warning: the product stays on the Gram-form kernels
        assert B.gram_mode(xp, xp) == 2
    sv = xp.sorted_view()
    assert xp.zmax2 > 20_000.0                              # (max |z1| + max |z2|)^2 > 80 000: the old f16-range limit of the split norms was 60 000
    assert sv.n_block - sv.n_compact > 0.25 * n, (sv.n_compact, sv.n_block)      # there ARE medium rows ...
    assert sv.n_compact % 512 == 0 and sv.n_block % 128 == 0
    g = torch.Generator().manual_seed(11)
    rows = torch.cat([sv.perm[:64].cpu(), sv.perm[sv.n_compact : sv.n_compact + 256].cpu(), sv.perm[sv.n_block - 128 : sv.n_block].cpu(), sv.perm[-64:].cpu(),
                      torch.randint(0, n, (200,), generator=g)]).unique()
    Krows = OK.kernel_matrix(kind, X[rows].double(), X.double(), ls, 1.0, x1_eq_x2=False, direct=True)
    for t in (1, 4, 11, 17, 33, 65, 70):
        V = torch.randn(n, t, generator=torch.Generator().manual_seed(t))
        out_t = B.kv(xp, xp, B.to_probe_major(V.to(dev)))
        got = out_t[:, rows.to(dev)].t().double().cpu()
        ref = Krows @ V.double()
        assert rel_err(got, ref) < 2e-5, (kind, t, rel_err(got, ref))
    # the entries themselves on medium rows
    cols = rows[:48]
    E = torch.zeros(n, cols.numel())
    E[cols, torch.arange(cols.numel())] = 1.0
    got = B.kv(xp, xp, B.to_probe_major(E.to(dev)))[:, rows.to(dev)].t().double().cpu()
    assert float((got - Krows[:, cols]).abs().max()) < 2e-5
    # backward sums against float64 autograd on the prepared coordinates (n small enough for dense float64 on the device)
    nb, t = 6144, 12
    Xb = X[:nb].to(dev)
    xb = B.prep_points(kind, Xb, torch.tensor([ls * 5.0]), Xb.mean(0))          # (a fifth of the points at five times the lengthscale: the same spacing)
    assert B.gram_mode(xb, xb) == 2 and xb.sorted_view().n_block - xb.sorted_view().n_compact > 0.25 * nb
    Lv = torch.randn(nb, t, generator=g, dtype=torch.float64).abs()
    Rv = torch.randn(nb, t, generator=g, dtype=torch.float64).abs()
    out, _ = B.kv_grad2(xb, xb, B.to_probe_major(Lv.float().to(dev)), B.to_probe_major(Rv.float().to(dev)), iso=True)
    z = xb.xp[:, :3].double()
    S = (z.unsqueeze(1) - z.unsqueeze(0)).pow(2).sum(-1)
    if kind == "rbf":
        Kd, dK = torch.exp2(-S), -math.log(2.0) * torch.exp2(-S)
    else:
        r = S.sqrt()
        Kd, dK = (1.0 + r + S / 3.0) * torch.exp(-r), -(1.0 + r) * torch.exp(-r) / 6.0
    W = Lv.to(dev) @ Rv.to(dev).t()
    tot, gs = float((W * Kd).sum()), float((W * dK * S).sum())
    assert abs(float(out[0]) - tot) < 2e-5 * abs(tot)
    assert abs(float(out[1]) - gs) < 5e-5 * abs(gs)          # sum W dk/ds s: the single-lengthscale sum (MODE 0 convention: out[1])


@pytest.mark.parametrize("split", [False, True], ids=["f32mfma", "split"])
def test_a_few_medium_and_wide_rows_share_one_direct_launch(split, dev, monkeypatch):
    """Round 6.  A cloud with a FEW medium and a FEW wide rows (the protein-shaped workload: 2048 + 232 of 36 584): ``kv_partials_sorted`` sends both
    to ONE direct-difference region launch (``REGION_MERGE_MAX_ROWS``) instead of two -- at that size each region's five launches are what the
    product costs.  Same numbers against float64 rows as the three-region form (forced by raising / zeroing the limit)."""
    from gpytorch_amd import backend as B

    monkeypatch.setattr(B, "SPLIT_CONTRACTION", split)
    n, d, ls = 24_000, 6, 0.57
    g = torch.Generator().manual_seed(5)
    X = torch.randn(n, d, generator=g).clamp_(-3.0, 3.0)                  # standardised features: a compact bulk, a sparse shell
    Xd = X.to(dev)
    xp = B.prep_points("rbf", Xd, torch.tensor([ls]), Xd.mean(0))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert B.gram_mode(xp, xp) == 2
    sv = xp.sorted_view()
    n_med, n_wide = sv.n_block - sv.n_compact, n - sv.n_block
    assert n_med > 0 and n_wide > 0 and n_med + n_wide <= B.REGION_MERGE_MAX_ROWS, (sv.n_compact, sv.n_block)
    rows = torch.cat([sv.perm[:64].cpu(), sv.perm[sv.n_compact : sv.n_compact + 128].cpu(), sv.perm[-128:].cpu()]).unique()
    Krows = OK.kernel_matrix("rbf", X[rows].double(), X.double(), ls, 1.0, x1_eq_x2=False, direct=True)
    for t in (1, 11, 65):
        V = torch.randn(n, t, generator=torch.Generator().manual_seed(t))
        vt = B.to_probe_major(V.to(dev))
        ref = Krows @ V.double()
        merged = B.kv(xp, xp, vt)[:, rows.to(dev)].t().double().cpu()
        monkeypatch.setattr(B, "REGION_MERGE_MAX_ROWS", 0)
        three = B.kv(xp, xp, vt)[:, rows.to(dev)].t().double().cpu()
        monkeypatch.setattr(B, "REGION_MERGE_MAX_ROWS", 4096)
        assert rel_err(merged, ref) < 2e-5 and rel_err(three, ref) < 2e-5, (t, rel_err(merged, ref), rel_err(three, ref))


@pytest.mark.parametrize("split", [False, True], ids=["f32mfma", "split"])
def test_region_launches_beside_the_compact_launch_are_bitwise_the_single_stream_product(split, dev, monkeypatch):
    """Round 6.  The medium / wide region launches of a block-centred product run on a second HIP stream beside the compact launch
    (``backend.REGION_STREAMS``: fork at the start of the product, join before the slabs are read).  Same kernels, same slabs, same order of the
    slab sums: the product must be BITWISE the one-stream product -- back-to-back products without a host sync in between (the region workspaces
    and the slabs are re-used by consecutive products: mBCG's pattern), three- and two-region forms, 1 ... 65 columns."""
    from gpytorch_amd import backend as B

    monkeypatch.setattr(B, "SPLIT_CONTRACTION", split)
    n, d, ls = 24_000, 6, 0.57
    g = torch.Generator().manual_seed(5)
    X = torch.randn(n, d, generator=g).clamp_(-3.0, 3.0)
    Xd = X.to(dev)
    xp = B.prep_points("rbf", Xd, torch.tensor([ls]), Xd.mean(0))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert B.gram_mode(xp, xp) == 2
    sv = xp.sorted_view()
    assert sv.n_compact > 0 and sv.n_compact < n
    for merge in (4096, 0):
        monkeypatch.setattr(B, "REGION_MERGE_MAX_ROWS", merge)
        for t in (1, 4, 11, 33, 65):
            vts = [B.to_probe_major(torch.randn(n, t, generator=torch.Generator().manual_seed(100 * t + k)).to(dev)) for k in range(6)]
            monkeypatch.setattr(B, "REGION_STREAMS", False)
            ref = [B.kv(xp, xp, v).clone() for v in vts]
            monkeypatch.setattr(B, "REGION_STREAMS", True)
            for rep in range(3):
                got = [B.kv(xp, xp, v) for v in vts]          # six products in flight, no host sync
                for a, b in zip(got, ref):
                    assert torch.equal(a, b), (merge, t, rep, float((a - b).abs().max()))


@pytest.mark.parametrize("kind", ["matern32", "matern52"])
@pytest.mark.parametrize("split", [False, True], ids=["f32mfma", "split"])
def test_contracted_points_near_the_extent_limit_of_the_saturating_norms(kind, split, dev, monkeypatch):
    """The block-centred Gram expansion saturates the split norm of a CONTRACTED point at 60 000 (f16 range) and relies on k being zero to
    float32 precision long before (gram_f16.hpp; include/gpamd.h: (max |z1| + max |z2|)^2 <= 2.5e7 is a HARD precondition of GPAMD_KV_GRAM with
    block centres -- the kernels do not check it, ``backend.gram_mode`` does).  Advisor finding (round 5): nothing exercised it near the limit.
    Here: 48 tight clusters scattered over a cube of half-width 1300 in the prepared (scaled) units -- |z_j - c| up to ~4500 between a row block's
    centre and a contracted point, extent (2 max |z|)^2 ~ 2e7 -- so that almost every pair is far beyond the saturation point while pairs inside a
    cluster carry the whole product.  Against the float64 oracle (reference formulas)."""
    from gpytorch_amd import backend as B

    monkeypatch.setattr(B, "SPLIT_CONTRACTION", split)
    n, d, ls = 24_576, 3, 1.0
    g = torch.Generator().manual_seed(5)
    centres = (torch.rand(48, d, generator=g) * 2 - 1) * 1300.0 / math.sqrt(3.0 if kind == "matern32" else 5.0)
    X = (centres[torch.randint(0, 48, (n,), generator=g)] + 0.6 * torch.randn(n, d, generator=g)).float()
    Xd = X.to(dev)
    xp = B.prep_points(kind, Xd, torch.tensor([ls]), Xd.mean(0))
    zmax = math.sqrt(xp.zmax2)
    assert 1.0e7 < (2 * zmax) ** 2 <= B.GRAM_MAX_EXTENT_SQ, (2 * zmax) ** 2      # close to the documented limit, inside it
    assert B.gram_mode(xp, xp) == 2
    rows = torch.cat([torch.arange(128), torch.randint(128, n - 128, (256,), generator=g), torch.arange(n - 128, n)]).unique()
    Krows = OK.kernel_matrix(kind, X[rows].double(), X.double(), ls, 1.0, x1_eq_x2=False, direct=True)
    for t in (1, 8, 33, 65):
        V = torch.randn(n, t, generator=torch.Generator().manual_seed(t))
        got = B.kv(xp, xp, B.to_probe_major(V.to(dev)))[:, rows.to(dev)].t().double().cpu()
        ref = Krows @ V.double()
        # (5e-5, not the 2e-5 of compact clouds: at |z| ~ 1300 the float32 PREPARED coordinates resolve 1.2e-4, so distances of order one inside a
        # cluster carry 1e-4 relative before any kernel arithmetic -- measured 3.4e-5 (Matern-3/2, one column), the same on both contractions)
        assert rel_err(got, ref) < 5e-5, (kind, t, rel_err(got, ref))
