"""GPU parity: SURVEY.md 8(f) rank 3 -- other stationary kernels and kernel composition on the fused path.

  * PeriodicKernel values / diagonal / ARD            gpytorch/kernels/periodic_kernel.py:125-142 (KeOps twin: kernels/keops/periodic_kernel.py)
  * ProductKernel of squared-exponential members      gpytorch/kernels/kernel.py:634-688 -- one fused operator over concatenated features
  * AdditiveKernel                                     gpytorch/kernels/kernel.py:592-632 -- matrix-free sum, mBCG / SLQ / backward per member
Ground truth: float64 restatement of the reference formulas + torch autograd (dense Cholesky for the MLL).
"""
import math

import pytest
import torch

from oracle import exact_gp as OG
from oracle import kernels as OK
from tests.util import make_data, rel_err

pytestmark = pytest.mark.gpu


periodic_ref = OK.periodic   # pinned to the reference's own PeriodicKernel.forward by tests/test_oracle_golden.py::test_periodic_and_rq_golden


@pytest.mark.parametrize("name", list("pqrs"))
def test_periodic_and_rq_against_reference_generated_fixtures(name, dev):
    """The product kernels against OUTPUTS OF THE REFERENCE'S OWN CODE (tests/golden/composite_values.npz: PeriodicKernel.forward,
    periodic_kernel.py:125-142, and RQKernel.forward, rq_kernel.py:61-74, executed through the reference's Kernel.covar_dist by
    tests/golden/make_golden.py): dense values, diagonal, and the gradients of sum(W * K) w.r.t. every hyper-parameter."""
    import os

    import numpy as np

    import gpytorch_amd as g

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "composite_values.npz"))
    x1 = torch.from_numpy(z[f"{name}_x1"]).float().to(dev)
    x2 = torch.from_numpy(z[f"{name}_x2"]).float().to(dev)
    W = torch.from_numpy(z[f"{name}_W"]).to(dev)
    ls, per, al = (torch.from_numpy(z[f"{name}_{k}"]).float() for k in ("ls", "period", "alpha"))
    d = x1.shape[1]
    ard = d if ls.shape[-1] > 1 else None
    eye = torch.eye(x2.shape[0], device=dev)

    def sig(raw):   # softplus chain rule: d softplus(raw) / d raw
        return torch.sigmoid(raw.detach().double().cpu())

    def close(got, want, raw, tol=3e-3):
        want = torch.from_numpy(np.asarray(want)).double().reshape(got.shape) * sig(raw).reshape(got.shape)
        return torch.allclose(got.double().cpu(), want, rtol=tol, atol=tol * float(want.abs().max()))

    kp = g.kernels.PeriodicKernel(ard_num_dims=ard).to(dev)
    kp.lengthscale, kp.period_length = ls, per
    assert rel_err(kp(x1, x2).to_dense(), torch.from_numpy(z[f"{name}_periodic"]).double()) < 1e-5
    assert torch.equal(kp(x1, diag=True).cpu().double(), torch.from_numpy(z[f"{name}_periodic_diag"]).double())
    ((kp(x1, x2) @ eye) * W.float()).sum().backward()
    assert close(kp.raw_lengthscale.grad, z[f"{name}_periodic_dls"], kp.raw_lengthscale)
    assert close(kp.raw_period_length.grad, z[f"{name}_periodic_dperiod"], kp.raw_period_length)

    kr = g.kernels.RQKernel(ard_num_dims=ard).to(dev)
    kr.lengthscale, kr.alpha = ls, al
    assert rel_err(kr(x1, x2).to_dense(), torch.from_numpy(z[f"{name}_rq"]).double()) < 1e-5
    assert torch.equal(kr(x1, diag=True).cpu().double(), torch.from_numpy(z[f"{name}_rq_diag"]).double())
    ((kr(x1, x2) @ eye) * W.float()).sum().backward()
    assert close(kr.raw_lengthscale.grad, z[f"{name}_rq_dls"], kr.raw_lengthscale)
    assert close(kr.raw_alpha.grad, z[f"{name}_rq_dalpha"], kr.raw_alpha)


@pytest.mark.parametrize("ard", [False, True])
def test_periodic_kernel_values_matmul_and_grads(ard, dev):
    import gpytorch_amd as g

    gen = torch.Generator().manual_seed(0)
    n, m, d = 300, 210, 3
    x1 = torch.rand(n, d, generator=gen) * 3
    x2 = torch.rand(m, d, generator=gen) * 3
    kern = g.kernels.PeriodicKernel(ard_num_dims=d if ard else None).to(dev)
    ls = torch.tensor([[0.9, 1.4, 2.0]]) if ard else torch.tensor([[1.3]])
    per = torch.tensor([[1.1, 0.7, 1.9]]) if ard else torch.tensor([[0.8]])
    kern.lengthscale = ls
    kern.period_length = per
    K = kern(x1.to(dev), x2.to(dev)).to_dense()
    Kref = periodic_ref(x1.double(), x2.double(), ls.double(), per.double())
    assert rel_err(K, Kref) < 1e-5
    assert torch.equal(kern(x1.to(dev), diag=True).cpu(), torch.ones(n))
    # matrix-free product and hyper-parameter gradients through the fused input gradient
    V = torch.randn(n, 7, generator=gen)
    op = kern(x1.to(dev))
    out = op @ V.to(dev)
    out.pow(2).sum().backward()
    ls64 = ls.double().requires_grad_(True)
    p64 = per.double().requires_grad_(True)
    ref = periodic_ref(x1.double(), x1.double(), ls64, p64) @ V.double()
    assert rel_err(out, ref) < 2e-5
    ref.pow(2).sum().backward()
    sig = lambda raw: torch.sigmoid(raw.detach().double().cpu())  # noqa: E731  (softplus chain rule)
    assert torch.allclose(kern.raw_lengthscale.grad.double().cpu(), ls64.grad * sig(kern.raw_lengthscale), rtol=2e-3, atol=1e-6 * float(ls64.grad.abs().max()))
    assert torch.allclose(kern.raw_period_length.grad.double().cpu(), p64.grad * sig(kern.raw_period_length), rtol=2e-3, atol=1e-6 * float(p64.grad.abs().max()))


def test_product_of_rbf_and_periodic_is_one_fused_operator(dev):
    import gpytorch_amd as g
    from gpytorch_amd.operators import FusedKernelLinearOperator

    gen = torch.Generator().manual_seed(1)
    x = torch.rand(400, 3, generator=gen) * 2
    k1 = g.kernels.ScaleKernel(g.kernels.RBFKernel(active_dims=[0, 2]))
    k2 = g.kernels.PeriodicKernel(active_dims=[1])
    kern = (k1 * k2).to(dev)
    k1.base_kernel.lengthscale = 0.6
    k1.outputscale = 1.7
    k2.lengthscale = 1.2
    k2.period_length = 0.5
    op = kern(x.to(dev))
    assert isinstance(op, FusedKernelLinearOperator)            # matrix-free: 2 + 2 feature dimensions
    xd = x.double()
    Kref = 1.7 * OK.rbf(xd[:, [0, 2]], xd[:, [0, 2]], 0.6, x1_eq_x2=True, direct=True) * periodic_ref(xd[:, [1]], xd[:, [1]], torch.tensor([[1.2]]), torch.tensor([[0.5]]))
    assert rel_err(op.to_dense(), Kref) < 1e-5
    V = torch.randn(400, 20, generator=gen)
    assert rel_err(op @ V.to(dev), Kref @ V.double()) < 2e-5
    # a product with a Matern member falls back to the dense elementwise product (the reference densifies as well)
    mixed = (g.kernels.RBFKernel() * g.kernels.MaternKernel(nu=1.5)).to(dev)
    Kd = mixed(x.to(dev)).to_dense()
    l0 = math.log(2.0)  # default raw parameter 0 -> softplus = ln 2
    assert rel_err(Kd, OK.rbf(xd, xd, l0, x1_eq_x2=True, direct=True) * OK.matern(xd, xd, l0, 1.5, x1_eq_x2=True, direct=True)) < 1e-5


@pytest.mark.parametrize("branch", ["cholesky", "bbmm"])
def test_additive_kernel_mll_and_grads(branch, dev):
    """ScaleKernel(RBF) + ScaleKernel(Matern-5/2): MLL value and every hyper-parameter gradient against dense float64 autograd
    (BBMM branch: probes shared with the oracle through deterministic_probes; log-det terms compared at the SLQ accuracy)."""
    import gpytorch_amd as g

    n, d = (500, 2) if branch == "cholesky" else (1800, 2)
    X, y = make_data(n, d)

    class M(g.models.ExactGP):
        def __init__(self, x, yy, lik):
            super().__init__(x, yy, lik)
            self.mean_module = g.means.ZeroMean()
            self.covar_module = g.kernels.ScaleKernel(g.kernels.RBFKernel()) + g.kernels.ScaleKernel(g.kernels.MaternKernel(nu=2.5))

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    lik = g.likelihoods.GaussianLikelihood().to(dev)
    m = M(X.float().to(dev), y.float().to(dev), lik).to(dev)
    ka, kb = m.covar_module.kernels
    ka.base_kernel.lengthscale, ka.outputscale = 0.25, 1.1
    kb.base_kernel.lengthscale, kb.outputscale = 0.9, 0.4
    lik.noise = 0.1
    mll = g.ExactMarginalLogLikelihood(lik, m)
    m.train()
    lik.train()
    S = g.settings
    with S.max_cholesky_size(10_000 if branch == "cholesky" else 0), S.cg_tolerance(1e-5), S.num_trace_samples(400), S.max_preconditioner_size(0):
        torch.manual_seed(0)
        val = mll(m(m.train_inputs[0]), m.train_targets)
        val.backward()
    p = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in (0.25, 1.1, 0.9, 0.4, 0.1)]
    Kh = p[1] * OK.rbf(X, X, p[0], x1_eq_x2=True, direct=True) + p[3] * OK.matern(X, X, p[2], 2.5, x1_eq_x2=True, direct=True) + p[4] * torch.eye(n, dtype=torch.float64)
    ref = OG.dense_log_prob(Kh, y) / n
    gref = torch.autograd.grad(ref, p)
    tol_v, tol_g = (2e-4, 3e-3) if branch == "cholesky" else (5e-3, 0.12)
    assert abs(float(val) - float(ref)) < tol_v * max(1.0, abs(float(ref)))
    sp = lambda v: 1.0 - math.exp(-v)  # noqa: E731
    got = [ka.base_kernel.raw_lengthscale.grad, ka.raw_outputscale.grad, kb.base_kernel.raw_lengthscale.grad, kb.raw_outputscale.grad, lik.noise_covar.raw_noise.grad]
    chain = [sp(0.25), sp(1.1), sp(0.9), sp(0.4), sp(0.1 - 1e-4)]
    assert all(gg is not None for gg in got)
    gv = torch.tensor([float(gg.sum()) for gg in got], dtype=torch.float64)
    wv = torch.tensor([float(rr) * cc for rr, cc in zip(gref, chain)], dtype=torch.float64)
    # (the gradient as a vector: single components can sit near zero where the stochastic trace estimate dominates them)
    assert float((gv - wv).norm() / wv.norm()) < tol_g, (gv, wv)


def test_additive_kernel_posterior_with_cg(dev):
    """Predictive mean / variance of an additive-kernel GP with CG + Lanczos forced, against the dense float64 posterior."""
    import gpytorch_amd as g

    n, ns, d = 1500, 100, 2
    X, y = make_data(n + ns, d)
    Xt, yt, Xs = X[:n], y[:n], X[n:]

    class M(g.models.ExactGP):
        def __init__(self, x, yy, lik):
            super().__init__(x, yy, lik)
            self.mean_module = g.means.ZeroMean()
            self.covar_module = g.kernels.ScaleKernel(g.kernels.RBFKernel()) + g.kernels.ScaleKernel(g.kernels.PeriodicKernel())

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    lik = g.likelihoods.GaussianLikelihood().to(dev)
    m = M(Xt.float().to(dev), yt.float().to(dev), lik).to(dev)
    ka, kb = m.covar_module.kernels
    ka.base_kernel.lengthscale, ka.outputscale = 0.3, 1.0
    kb.base_kernel.lengthscale, kb.base_kernel.period_length, kb.outputscale = 1.5, 0.7, 0.5
    lik.noise = 0.2
    m.eval()
    lik.eval()
    S = g.settings
    with torch.no_grad(), S.max_cholesky_size(0), S.eval_cg_tolerance(1e-5):
        pred = m(Xs.float().to(dev))
        mu, var = pred.mean, pred.variance

    def kfun(a, b, same):
        return 1.0 * OK.rbf(a, b, 0.3, x1_eq_x2=same, direct=True) + 0.5 * periodic_ref(a, b, torch.tensor([[1.5]]), torch.tensor([[0.7]]))

    Kh = kfun(Xt, Xt, True) + 0.2 * torch.eye(n, dtype=torch.float64)
    Ks = kfun(Xs, Xt, False)
    Lc = torch.linalg.cholesky(Kh)
    mu_ref = Ks @ torch.cholesky_solve(yt.unsqueeze(-1), Lc).squeeze(-1)
    w = torch.linalg.solve_triangular(Lc, Ks.t(), upper=False)
    var_ref = 1.5 - w.pow(2).sum(0)
    assert rel_err(mu, mu_ref) < 1e-3
    assert float((var.double().cpu() - var_ref).abs().max()) < 2e-3


@pytest.mark.parametrize("ard", [False, True])
def test_rq_kernel_values_products_and_all_gradients(ard, dev):
    """RQKernel (gpytorch/kernels/rq_kernel.py:60-74) as a native covariance family: dense values, K @ V on every column-count
    kernel, and gradients w.r.t. lengthscale(s), alpha and the inputs against float64 autograd."""
    import gpytorch_amd as g

    gen = torch.Generator().manual_seed(0)
    n, m, d = 700, 450, 3
    x1 = torch.rand(n, d, generator=gen)
    x2 = torch.rand(m, d, generator=gen)
    kern = g.kernels.RQKernel(ard_num_dims=d if ard else None).to(dev)
    ls = torch.tensor([[0.5, 0.8, 0.35]]) if ard else torch.tensor([[0.45]])
    kern.lengthscale = ls
    kern.alpha = 1.7
    Kref = OK.rq(x1.double(), x2.double(), ls.double(), 1.7, x1_eq_x2=False, direct=True)
    assert rel_err(kern(x1.to(dev), x2.to(dev)).to_dense(), Kref) < 1e-5
    for t in (1, 4, 11, 16, 40, 65):
        V = torch.randn(m, t, generator=gen)
        assert rel_err(kern(x1.to(dev), x2.to(dev)) @ V.to(dev), Kref @ V.double()) < 3e-5, t
    xa = x1.to(dev).requires_grad_(True)
    V = torch.randn(n, 9, generator=gen)
    out = kern(xa, xa) @ V.to(dev)
    out.pow(2).sum().backward()
    ls64 = ls.double().requires_grad_(True)
    a64 = torch.tensor(1.7, dtype=torch.float64, requires_grad=True)
    x64 = x1.double().requires_grad_(True)
    ref = OK.rq(x64, x64, ls64, a64, x1_eq_x2=False, direct=True) @ V.double()
    ref.pow(2).sum().backward()
    sig = lambda raw: torch.sigmoid(raw.detach().double().cpu())  # noqa: E731
    assert torch.allclose(kern.raw_lengthscale.grad.double().cpu(), ls64.grad * sig(kern.raw_lengthscale), rtol=2e-3)
    assert abs(float(kern.raw_alpha.grad) - float(a64.grad) * float(sig(kern.raw_alpha))) < 2e-3 * abs(float(a64.grad))
    assert rel_err(xa.grad, x64.grad) < 2e-3


def test_rq_gp_mll_bbmm_and_cholesky(dev):
    """ScaleKernel(RQKernel) ExactGP: MLL value on the Cholesky and BBMM branches and the alpha gradient (Cholesky branch)
    against dense float64 autograd."""
    import gpytorch_amd as g

    n, d = 1200, 2
    X, y = make_data(n, d)

    class M(g.models.ExactGP):
        def __init__(self, x, yy, lik):
            super().__init__(x, yy, lik)
            self.mean_module = g.means.ZeroMean()
            self.covar_module = g.kernels.ScaleKernel(g.kernels.RQKernel())

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    p = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in (0.3, 2.2, 1.3, 0.1)]   # ls, alpha, outputscale, noise
    Kh = p[2] * OK.rq(X, X, p[0], p[1], x1_eq_x2=True, direct=True) + p[3] * torch.eye(n, dtype=torch.float64)
    ref = OG.dense_log_prob(Kh, y) / n
    gref = torch.autograd.grad(ref, p)
    for branch in ("cholesky", "bbmm"):
        lik = g.likelihoods.GaussianLikelihood().to(dev)
        m = M(X.float().to(dev), y.float().to(dev), lik).to(dev)
        m.covar_module.base_kernel.lengthscale = 0.3
        m.covar_module.base_kernel.alpha = 2.2
        m.covar_module.outputscale = 1.3
        lik.noise = 0.1
        mll = g.ExactMarginalLogLikelihood(lik, m)
        m.train()
        lik.train()
        S = g.settings
        with S.max_cholesky_size(10_000 if branch == "cholesky" else 0), S.cg_tolerance(1e-5), S.num_trace_samples(300), S.max_preconditioner_size(0):
            torch.manual_seed(0)
            val = mll(m(m.train_inputs[0]), m.train_targets)
            val.backward()
        tol_v, tol_g = (2e-4, 3e-3) if branch == "cholesky" else (5e-3, 0.15)
        assert abs(float(val) - float(ref)) < tol_v * max(1.0, abs(float(ref))), branch
        sp = lambda v: 1.0 - math.exp(-v)  # noqa: E731
        got = torch.tensor([float(m.covar_module.base_kernel.raw_lengthscale.grad.sum()), float(m.covar_module.base_kernel.raw_alpha.grad.sum()),
                            float(m.covar_module.raw_outputscale.grad), float(lik.noise_covar.raw_noise.grad.sum())], dtype=torch.float64)
        want = torch.tensor([float(gref[0]) * sp(0.3), float(gref[1]) * sp(2.2), float(gref[2]) * sp(1.3), float(gref[3]) * sp(0.1 - 1e-4)], dtype=torch.float64)
        assert float((got - want).norm() / want.norm()) < tol_g, (branch, got, want)


def test_rhs_refinement_on_the_additive_operator(dev):
    """Round 6: ``settings.rhs_refinement`` on the SUM operator (round 5 had it on the single-kernel operator, round 6 on the Kronecker one): the
    float32 mBCG solve of an ill-conditioned additive system is refined with a float64 residual through the members' fused float64 products, and
    the exact predictive variance of f takes the variational quadratic form (one float64 product, no second solve).  Against dense float64."""
    import gpytorch_amd as g

    n, ns, d = 4000, 64, 2
    X, y = make_data(n + ns, d)
    X, y = X.float().double(), y.float().double()                  # the model's float32 inputs, exactly
    Xt, yt, Xs = X[:n], y[:n], X[n:]

    class M(g.models.ExactGP):
        def __init__(self, x, yy, lik):
            super().__init__(x, yy, lik)
            self.mean_module = g.means.ZeroMean()
            self.covar_module = g.kernels.ScaleKernel(g.kernels.RBFKernel()) + g.kernels.ScaleKernel(g.kernels.MaternKernel(nu=2.5))

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    lik = g.likelihoods.GaussianLikelihood().to(dev)
    m = M(Xt.float().to(dev), yt.float().to(dev), lik).to(dev)
    ka, kb = m.covar_module.kernels
    ka.base_kernel.lengthscale, ka.outputscale = 0.3, 1.0
    kb.base_kernel.lengthscale, kb.outputscale = 0.8, 0.5
    lik.noise = 1e-3                                               # kappa ~ 1e6: float32 solves stall near 1e-3
    Kh = 1.0 * OK.rbf(Xt, Xt, 0.3, x1_eq_x2=True, direct=True) + 0.5 * OK.matern(Xt, Xt, 0.8, 2.5, x1_eq_x2=True, direct=True) + 1e-3 * torch.eye(n, dtype=torch.float64)
    Ks = 1.0 * OK.rbf(Xs, Xt, 0.3, x1_eq_x2=False, direct=True) + 0.5 * OK.matern(Xs, Xt, 0.8, 2.5, x1_eq_x2=False, direct=True)
    Lc = torch.linalg.cholesky(Kh)
    sol_ref = torch.cholesky_solve(yt.unsqueeze(-1), Lc)
    fvar_ref = 1.5 - (Ks * torch.cholesky_solve(Ks.t(), Lc).t()).sum(-1)
    S = g.settings
    op = lik(m.train()(Xt.float().to(dev))).lazy_covariance_matrix
    assert op.float64_product_available()
    V = torch.randn(n, 3, generator=torch.Generator().manual_seed(2), dtype=torch.float64)
    assert rel_err(op.matmul_float64(V.to(dev)), Kh @ V) < 1e-5      # (float64 arithmetic on the float32-PREPARED points: their rounding, 1e-7 in z, is what is left)
    errs = {}
    for refine in (False, True):
        with torch.no_grad(), S.max_cholesky_size(0), S.cg_tolerance(1e-4), S.max_cg_iterations(4000), S.rhs_refinement(refine):
            errs[refine] = rel_err(op.solve(yt.float().to(dev).unsqueeze(-1)), sol_ref)
    assert errs[True] < 0.2 * errs[False] and errs[True] < 1e-4, errs
    m.eval(), lik.eval()
    verr = {}
    for refine in (False, True):
        m.train(), m.eval()
        with torch.no_grad(), S.max_cholesky_size(0), S.eval_cg_tolerance(1e-4), S.cg_tolerance(1e-4), S.max_cg_iterations(4000), S.rhs_refinement(refine):
            fvar = m(Xs.float().to(dev)).variance.double().cpu()
        verr[refine] = float((fvar - fvar_ref).abs().max() / 1e-3)      # in units of the noise
    assert verr[True] < 0.05 and verr[True] < verr[False], verr


def test_sum_and_product_known_answers_of_the_reference(dev):
    """test/kernels/test_additive_and_product_kernels.py:33-157 -- dense values, batch members, and ``diag=True`` on two DIFFERENT inputs (the
    product of squared-exponential members used to answer with the all-ones diagonal of K(x, x) there)."""
    import gpytorch_amd as g
    from tests.known_answers import check_sum_product_known_answers

    check_sum_product_known_answers(g, dev)


def test_stationary_and_periodic_unit_tests_of_the_reference(dev):
    """test/kernels/test_rbf_kernel.py:20-125, test/kernels/test_periodic_kernel.py:20-88 (closed forms; ``last_dim_is_batch`` with ARD lengthscales,
    re-viewed hyper-parameter shapes)."""
    import gpytorch_amd as g
    from tests.known_answers import check_stationary_and_periodic_unit_tests

    check_stationary_and_periodic_unit_tests(g, dev)


@pytest.mark.parametrize("family", ["rbf", "matern32", "matern12", "matern52", "periodic", "rq", "scale_rbf"])
def test_generic_kernel_battery_of_the_reference(family, dev):
    """gpytorch/test/base_kernel_test_case.py:30-197: active dimensions, batch inputs under kernels with and without a batch shape, ARD, ``diag=True``,
    kernel ``__getitem__`` / ``expand_batch``, pickling -- over every kernel family of the path."""
    import gpytorch_amd as g
    from tests.kernel_battery import families, run_battery

    (_, make, make_ard), = [f for f in families(g) if f[0] == family]
    run_battery(make, make_ard, dev)


def test_scale_kernel_unit_tests_of_the_reference(dev):
    """test/kernels/test_scale_kernel.py:24-127 (tests/known_answers.py)."""
    import gpytorch_amd as g
    from tests.known_answers import check_scale_kernel_unit_tests

    check_scale_kernel_unit_tests(g, dev)
