"""GPU: far-pair TILE CULLING of the fused products (``settings.far_pair_cutoff``, ``gpamd_kv_partials_far_f32``, csrc/kv_cull.hpp).

Opt-in: the reference evaluates every pair (its KeOps seam reduces over all j, ``gpytorch/kernels/keops/rbf_kernel.py:44-55``) and so does the
library by default.  With a cutoff eps, a 128-point tile of the contracted cloud is skipped for a block of output rows when their bounding spheres
are so far apart that every covariance between them is <= eps.  What is asserted, against the float64 oracle (every pair evaluated):

  * the stated bound: |culled - exact|_ic <= eps * sum_j |V_jc| on top of the usual 2e-5 relative accuracy of the fused kernels;
  * that tiles WERE dropped: with a large eps the culled product differs from the exact one by more than float32 noise (and still obeys the bound);
  * eps = 1e-7 is invisible at the kernels' own accuracy -- the product, an mBCG solve, the MLL and the posterior through the model API agree with the
    un-culled run;
  * rectangular products (test points against training points) and every generation mode the culling rides on (block-centred Gram form with compact /
    medium / wide rows, direct differences: Matern-1/2).
"""
import math
import warnings

import pytest
import torch

from oracle import kernels as OK
from tests.util import rel_err

pytestmark = pytest.mark.gpu


def road_like(n, seed=0):
    """Points along random smooth planar curves + a slowly varying third coordinate, z-scored (the 3droad stand-in of scripts/reference_workloads.py
    in small: a locally one-dimensional cloud many lengthscales wide)."""
    g = torch.Generator().manual_seed(seed)
    roads = 40
    per = (n + roads - 1) // roads
    t = torch.linspace(0, 1, per).unsqueeze(0)
    p0 = torch.rand(roads, 1, 2, generator=g) * 10.0
    ang = torch.rand(roads, 1, generator=g) * 2 * math.pi
    curv = (torch.rand(roads, 1, generator=g) - 0.5) * 6.0
    length = 0.5 + 2.5 * torch.rand(roads, 1, generator=g)
    th = ang + curv * t
    step = length / per
    xy = (p0 + torch.stack([torch.cumsum(torch.cos(th) * step, 1), torch.cumsum(torch.sin(th) * step, 1)], -1)).reshape(-1, 2)[:n]
    xy = xy + 0.002 * torch.randn(xy.shape, generator=g)
    third = torch.sin(0.7 * xy[:, 0]) * torch.cos(0.5 * xy[:, 1]) + 0.05 * torch.randn(n, generator=g)
    X = torch.cat([xy, third.unsqueeze(-1)], -1)[torch.randperm(n, generator=g)]
    return ((X - X.mean(0, keepdim=True)) / (X.std(0, keepdim=True) + 1e-6)).float().contiguous()


def _prep(B, kind, X, ls, dev, shift=None):
    Xd = X.to(dev)
    return B.prep_points(kind, Xd, torch.tensor([ls]), Xd.mean(0) if shift is None else shift)


@pytest.mark.parametrize("kind,ls", [("rbf", 0.05), ("matern52", 0.05), ("matern32", 0.03), ("matern12", 0.02)])
def test_culled_product_obeys_its_bound_and_drops_tiles(kind, ls, dev):
    import gpytorch_amd as g
    from gpytorch_amd import backend as B

    n = 12_000
    X = road_like(n, seed=1)
    xp = _prep(B, kind, X, ls, dev)
    rows = torch.cat([torch.arange(150), torch.randint(150, n - 150, (400,), generator=torch.Generator().manual_seed(3)), torch.arange(n - 150, n)]).unique()
    Krows = OK.kernel_matrix(kind, X[rows].double(), X.double(), ls, 1.0, x1_eq_x2=False, direct=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # (Matern-1/2 and very wide clouds: the direct-difference fallback warning)
        for t in (8, 11, 33, 65):
            V = torch.randn(n, t, generator=torch.Generator().manual_seed(t))
            vt = B.to_probe_major(V.to(dev))
            ref = Krows @ V.double()
            exact = B.kv(xp, xp, vt)[:, rows.to(dev)].t().double().cpu()
            assert rel_err(exact, ref) < 2e-5
            l1 = V.double().abs().sum(0)                                   # sum_j |V_jc|
            # (a dropped tile lies the cutoff PLUS both sphere radii away: only coarse cutoffs -- k <= 0.5 for the steep RBF tail, 0.1 for the Matern
            # families -- make the culling visible in the VALUES of this cloud)
            for eps, must_differ in ((1e-7, False), (0.5 if kind == "rbf" else 0.1, True)):
                with g.settings.far_pair_cutoff(eps):
                    sq = B.far_cull(xp, xp)
                    assert sq is not None and B.far_kept_fraction(xp, xp, sq, 128) < 0.5
                    got = B.kv(xp, xp, vt)[:, rows.to(dev)].t().double().cpu()
                bound = eps * l1.unsqueeze(0) + 2e-5 * ref.abs().max()
                assert bool(((got - ref).abs() <= bound).all()), (kind, t, eps, float(((got - ref).abs() - bound).max()))
                if must_differ:
                    # tiles holding covariances up to eps were dropped: far beyond the float32 noise of the exact product
                    noise = float((exact - ref).abs().max())
                    assert float((got - exact).abs().max()) > max(5e-6 * float(ref.abs().max()), 3 * noise), (kind, t, noise)
                else:
                    assert rel_err(got, ref) < 2e-5


@pytest.mark.parametrize("few_below", [None, 1.01], ids=["few_by_policy", "few_forced"])
def test_rectangular_product_and_few_columns(few_below, dev, monkeypatch):
    """Test points against training points (two different sorted views); fewer than five columns: where few enough tiles survive (the policy
    ``FAR_FEW_MAX_KEPT``; forced in the second variant) ``kv_flags`` adds KV_SPLIT_FEW and they run culled on the split kernels as well (one mostly
    empty 32-column tile) instead of un-culled on the few-column kernels."""
    import gpytorch_amd as g
    from gpytorch_amd import backend as B

    if few_below is not None:
        monkeypatch.setattr(B, "FAR_FEW_MAX_KEPT", few_below)

    n, m, ls = 5000, 9000, 0.06
    X1, X2 = road_like(n, seed=4), road_like(m, seed=5)
    sh = X2.mean(0).to(dev)
    p1, p2 = _prep(B, "matern52", X1, ls, dev, sh), _prep(B, "matern52", X2, ls, dev, sh)
    K = OK.kernel_matrix("matern52", X1.double(), X2.double(), ls, 1.0, x1_eq_x2=False, direct=True)
    for t in (1, 3, 6, 40):
        V = torch.randn(m, t, generator=torch.Generator().manual_seed(t))
        vt = B.to_probe_major(V.to(dev))
        with g.settings.far_pair_cutoff(1e-7):
            sq = B.far_cull(p1, p2)
            assert sq is not None
            assert bool(B.kv_flags(p1, p2, t) & B.KV_SPLIT_FEW) == (t < 5 and B.far_kept_fraction(p1, p2, sq, 512) < B.FAR_FEW_MAX_KEPT)
            if few_below is not None:
                assert bool(B.kv_flags(p1, p2, t) & B.KV_SPLIT_FEW) == (t < 5)
            got = B.from_probe_major(B.kv(p1, p2, vt), n)
        assert not (B.kv_flags(p1, p2, t) & B.KV_SPLIT_FEW)
        assert rel_err(got, K @ V.double()) < 2e-5, t
    # the diagonal epilogue after the rows were taken back to the original order
    V = torch.randn(m, 7, generator=torch.Generator().manual_seed(11))
    vt = B.to_probe_major(V.to(dev))
    K2 = OK.kernel_matrix("matern52", X2.double(), X2.double(), ls, 1.0, x1_eq_x2=True, direct=True)
    with g.settings.far_pair_cutoff(1e-7):
        got = B.from_probe_major(B.kv(p2, p2, vt, scale=torch.tensor([1.3], device=dev), dscale=torch.tensor([0.2], device=dev), vd=vt), m)
    assert rel_err(got, 1.3 * (K2 @ V.double()) + 0.2 * V.double()) < 2e-5


def test_compact_clouds_and_small_problems_are_left_alone(dev):
    import gpytorch_amd as g
    from gpytorch_amd import backend as B

    X = torch.rand(6000, 3, generator=torch.Generator().manual_seed(0))
    with g.settings.far_pair_cutoff(1e-7):
        assert B.far_cull(_prep(B, "rbf", X, 0.5, dev), _prep(B, "rbf", X, 0.5, dev)) is None          # narrower than the cutoff: nothing is far
        assert B.far_cull(_prep(B, "rbf", X[:800], 0.01, dev), _prep(B, "rbf", X[:800], 0.01, dev)) is None   # launch-bound sizes
        xq = B.prep_points("rq", X.to(dev), torch.tensor([0.05]), None, 1.2)
        assert B.far_cull(xq, xq) is None                                                            # heavy tail: k never falls to 1e-7 inside the cloud
    xp = _prep(B, "rbf", X, 0.02, dev)
    assert B.far_cull(xp, xp) is None                                                                # the default: off


def test_solve_mll_and_posterior_through_the_model_api(dev):
    """mBCG (sorted search directions every iteration), the MLL with gradients and the posterior with LOVE, culled at eps = 1e-7 against the
    un-culled run on the same probes, and against dense float64."""
    import gpytorch_amd as g
    from gpytorch_amd import backend as B

    n, ls = 8192, 0.05
    X = road_like(n + 500, seed=7)
    Xtr, Xte = X[:n], X[n:]
    y = (torch.sin(3 * Xtr[:, 0]) + torch.cos(2 * Xtr[:, 1]) + 0.1 * torch.randn(n, generator=torch.Generator().manual_seed(1))).float()

    class M(g.models.ExactGP):
        def __init__(self, x, yy, lik):
            super().__init__(x, yy, lik)
            self.mean_module = g.means.ZeroMean()
            self.covar_module = g.kernels.ScaleKernel(g.kernels.MaternKernel(nu=2.5))

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    def run(eps):
        lik = g.likelihoods.GaussianLikelihood().to(dev)
        m = M(Xtr.to(dev), y.to(dev), lik).to(dev)
        m.covar_module.base_kernel.lengthscale, m.covar_module.outputscale, lik.noise = ls, 1.1, 0.05
        mll = g.ExactMarginalLogLikelihood(lik, m)
        S = g.settings
        with warnings.catch_warnings(), S.far_pair_cutoff(eps), S.max_cholesky_size(0), S.deterministic_probes(True):
            warnings.simplefilter("ignore")
            m.train(), lik.train()
            torch.manual_seed(0)                   # the same probe draw and the same Lanczos start vectors in both runs
            with S.cg_tolerance(1e-3), S.num_trace_samples(30):
                loss = -mll(m(Xtr.to(dev)), y.to(dev))
                loss.backward()
            grads = [p.grad.detach().clone().reshape(-1) for p in m.parameters()]
            m.eval(), lik.eval()
            torch.manual_seed(1)
            with torch.no_grad(), S.fast_pred_var(), S.eval_cg_tolerance(1e-4):
                pred = lik(m(Xte.to(dev)))
                mu, var = pred.mean.cpu(), pred.variance.cpu()
            S.deterministic_probes.reset()
        return float(loss.detach()), torch.cat(grads).cpu(), mu, var

    l0, g0, mu0, var0 = run(None)
    l1, g1, mu1, var1 = run(1e-7)
    assert abs(l1 - l0) < 1e-5 * max(1.0, abs(l0))
    assert rel_err(g1, g0) < 1e-4
    # (both posteriors stop their solves at eval_cg_tolerance = 1e-4 and build the LOVE root by Lanczos: products that differ at the 1e-6 level move
    # either by a few 1e-4 -- measured 2.8e-4 on the mean)
    assert rel_err(mu1, mu0) < 1e-3 and rel_err(var1, var0) < 1e-3
    # and the truth: dense float64 posterior mean
    Kh = 1.1 * OK.kernel_matrix("matern52", Xtr.double(), Xtr.double(), ls, 1.0, x1_eq_x2=True, direct=True) + 0.05 * torch.eye(n, dtype=torch.float64)
    Ks = 1.1 * OK.kernel_matrix("matern52", Xte.double(), Xtr.double(), ls, 1.0, x1_eq_x2=False, direct=True)
    assert rel_err(mu1, Ks @ torch.linalg.solve(Kh, y.double())) < 2e-3


@pytest.mark.parametrize("kind,ls,t", [("matern52", 0.05, 11), ("rbf", 0.05, 30), ("matern32", 0.1, 11), ("matern12", 0.03, 11), ("matern52", 0.012, 11)])
def test_bilinear_derivative_is_culled_too(kind, ls, t, dev):
    """The backward: the Gram-form derivative kernel (compact + medium rows), the direct-difference derivative of the wide rows and the all-direct
    case (Matern-1/2; a cloud outside every Gram policy) walk the same tile lists (``gpamd_kv_grad2_far_f32`` / ``gpamd_kv_grad_far_f32``).  Left / right
    vectors of one sign, so that the sums do not cancel and a relative comparison with the un-culled kernels means something: eps = 1e-7 is invisible
    (the sums are dominated by near pairs: even a coarse cutoff moves them by 1e-5 only, so that steps ARE skipped is read off the tile list itself),
    and the input gradients come back in the original row order."""
    import gpytorch_amd as g
    from gpytorch_amd import backend as B

    n = 20_000
    X = road_like(n, seed=9)
    xp = _prep(B, kind, X, ls, dev)
    gen = torch.Generator().manual_seed(t)
    lt = B.to_probe_major(torch.rand(n, t, generator=gen).to(dev) + 0.1)
    rt = B.to_probe_major(torch.rand(n, t, generator=gen).to(dev) + 0.1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gram = B.grad_gram_ok(xp, xp)

        def run(want_x):
            if gram:
                return B.kv_grad2(xp, xp, lt, rt, iso=True, want_gz1=want_x)
            return B.kv_grad(xp, xp, lt, rt, iso=True), None

        g0, gx0 = run(False)
        with g.settings.far_pair_cutoff(1e-7):
            assert B.far_cull(xp, xp) is not None
            g1, gx1 = run(False)
            # steps really are skipped: the tile list of unit 0 (the first 128 rows of the curve order against the first j chunk) as the list kernel
            # left it in the workspace -- ascending step starts, then the terminator (= the chunk length): far fewer entries than the chunk has steps
            tl = B._far_ws[torch.device(dev)][:8192].tolist() if torch.device(dev) in B._far_ws else B._far_ws[next(iter(B._far_ws))][:8192].tolist()
            k = 1
            while k < len(tl) and tl[k] > tl[k - 1]:
                k += 1
            chunk = tl[k - 1]
            assert chunk % 64 == 0 and 0 < chunk <= B.round_up(n, 64) and k - 1 < 0.7 * (chunk // 64), (kind, k, chunk)
        sc = float(g0.abs().max())
        assert float((g1 - g0).abs().max()) < 2e-5 * sc, (kind, g0.tolist(), g1.tolist())
        if gram and kind != "rq":
            _, gx0 = run(True)
            with g.settings.far_pair_cutoff(1e-7):
                _, gx1 = run(True)
            assert float((gx1 - gx0).abs().max()) < 2e-5 * float(gx0.abs().max())
