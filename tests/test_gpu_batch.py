"""GPU parity: batch dimensions on the kernel / operator seam.

  * the batched-matmul check of the fused-kernel harness: ``x1 = randn(3, 2, 100, 3)``, ``kern(x1, x1) @ randn(3, 2, 100, 1)``
    against the dense kernel, norm of the difference < 1e-3            gpytorch/test/base_keops_test_case.py:87-101
  * kernel with ``batch_shape`` (per-member lengthscales), batch getitem   test/lazy/test_lazy_evaluated_kernel_tensor.py:116-124
  * batch-GP marginal log likelihood == the per-member MLLs (and both == dense float64), batch-shaped hyper-parameter gradients
    test/mlls/test_exact_marginal_log_likelihood.py:52-69
  * train on a batch, test on a batch / on a single set, input gradients  test/examples/test_batch_gp_regression.py:124-172
"""
import math

import pytest
import torch

from oracle import exact_gp as OG
from oracle import kernels as OK
from tests.util import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["rbf", "matern52"])
def test_keops_harness_batch_matmul(kind, dev):
    import gpytorch_amd as g

    torch.manual_seed(0)
    x1 = torch.randn(3, 2, 100, 3)
    rhs = torch.randn(3, 2, 100, 1)
    kern = (g.kernels.RBFKernel() if kind == "rbf" else g.kernels.MaternKernel(nu=2.5)).to(dev)
    op = kern(x1.to(dev), x1.to(dev))
    assert op.shape == torch.Size([3, 2, 100, 100])
    res1 = op.matmul(rhs.to(dev))
    ls = float(kern.lengthscale)
    K = torch.stack([torch.stack([OK.kernel_matrix(kind, x1[i, j].double(), x1[i, j].double(), ls, 1.0, x1_eq_x2=True) for j in range(2)]) for i in range(3)])
    res2 = K @ rhs.double()
    assert float(torch.norm(res1.double().cpu() - res2)) < 1e-3
    assert rel_err(op.to_dense(), K) < 1e-5
    assert rel_err(op.diagonal(), K.diagonal(dim1=-2, dim2=-1)) < 1e-6


def test_kernel_batch_shape_and_getitem(dev):
    import gpytorch_amd as g

    torch.manual_seed(1)
    x1, x2 = torch.randn(5, 6), torch.randn(5, 6)
    kern = g.kernels.RBFKernel(batch_shape=torch.Size([2])).to(dev)
    kern.lengthscale = torch.tensor([[[1.5]], [[0.7]]])
    k = kern(x1.to(dev), x2.to(dev))
    assert k.size() == torch.Size([2, 5, 5])
    assert k[..., :4, :3].size() == torch.Size([2, 4, 3])
    for b, ls in enumerate((1.5, 0.7)):
        assert rel_err(k.to_dense()[b], OK.rbf(x1.double(), x2.double(), ls, x1_eq_x2=False)) < 1e-5
        assert rel_err(k[b].to_dense(), OK.rbf(x1.double(), x2.double(), ls, x1_eq_x2=False)) < 1e-5


def _batch_model(g, X, Y, batch_shape, dev):
    class M(g.models.ExactGP):
        def __init__(self, x, y, lik):
            super().__init__(x, y, lik)
            self.mean_module = g.means.ConstantMean(batch_shape=batch_shape)
            self.covar_module = g.kernels.ScaleKernel(g.kernels.RBFKernel(batch_shape=batch_shape), batch_shape=batch_shape)

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    lik = g.likelihoods.GaussianLikelihood(batch_shape=batch_shape).to(dev)
    return M(X.to(dev), Y.to(dev), lik).to(dev), lik


@pytest.mark.parametrize("branch", ["cholesky", "bbmm"])
def test_batch_gp_mll_equals_member_mlls(branch, dev):
    import gpytorch_amd as g

    n, d = 600, 2
    gen = torch.Generator().manual_seed(0)
    X = torch.rand(2, n, d, generator=gen)
    Y = torch.stack([torch.sin(4 * X[0, :, 0]) + 0.1 * torch.randn(n, generator=gen), torch.cos(3 * X[1].sum(-1)) + 0.1 * torch.randn(n, generator=gen)])
    m, lik = _batch_model(g, X, Y, torch.Size([2]), dev)
    ls, os_, nz = [0.3, 0.5], [1.2, 0.8], [0.1, 0.2]
    m.covar_module.base_kernel.lengthscale = torch.tensor(ls).view(2, 1, 1)
    m.covar_module.outputscale = torch.tensor(os_)
    lik.noise = torch.tensor(nz).view(2, 1)
    mll = g.ExactMarginalLogLikelihood(lik, m)
    m.train()
    lik.train()
    S = g.settings
    Z = torch.randn(n, 300, generator=gen)
    S.deterministic_probes.probe_vectors = Z.to(dev)
    try:
        with S.max_cholesky_size(10_000 if branch == "cholesky" else 0), S.deterministic_probes(True), S.cg_tolerance(1e-4), S.max_preconditioner_size(0):
            val = mll(m(m.train_inputs[0]), m.train_targets)
            assert val.shape == torch.Size([2])
            val.sum().backward()
    finally:
        S.deterministic_probes.probe_vectors = None
    g_ls = m.covar_module.base_kernel.raw_lengthscale.grad
    assert g_ls.shape == torch.Size([2, 1, 1]) and bool((g_ls != 0).all())
    for b in range(2):
        ref, gref = OG.dense_mll_and_grads("rbf", X[b].double(), Y[b].double(), ls[b], os_[b], nz[b])
        tol = 2e-4 if branch == "cholesky" else 2e-2   # BBMM: 300-probe SLQ log-det
        assert abs(float(val[b]) - float(ref)) < tol * max(1.0, abs(float(ref))), (b, float(val[b]), float(ref))
        if branch == "cholesky":
            chain = 1.0 - math.exp(-ls[b])
            assert abs(float(g_ls[b]) - float(gref[0]) * chain) < 2e-3 * abs(float(gref[0]) * chain) + 1e-6


def test_train_on_batch_test_on_batch(dev):
    """test_batch_gp_regression.py::test_train_on_batch_test_on_batch, CG forced for every solve / log-det."""
    import gpytorch_amd as g

    torch.manual_seed(0)
    train_x1 = torch.linspace(0, 2, 11).unsqueeze(-1)
    train_y1 = torch.sin(train_x1 * (2 * math.pi)).squeeze()
    train_y1 = train_y1 + torch.randn_like(train_y1) * 0.01
    test_x1 = torch.linspace(0, 2, 51).unsqueeze(-1)
    test_y1 = torch.sin(test_x1 * (2 * math.pi)).squeeze()
    train_x2 = torch.linspace(0, 1, 11).unsqueeze(-1)
    train_y2 = torch.sin(train_x2 * (2 * math.pi)).squeeze()
    train_y2 = train_y2 + torch.randn_like(train_y2) * 0.01
    test_x2 = torch.linspace(0, 1, 51).unsqueeze(-1)
    test_y2 = torch.sin(test_x2 * (2 * math.pi)).squeeze()
    train_x12 = torch.stack([train_x1, train_x2]).to(dev)
    train_y12 = torch.stack([train_y1, train_y2]).to(dev)
    test_x12 = torch.stack([test_x1, test_x2]).to(dev)

    class ExactGPModel(g.models.ExactGP):
        def __init__(self, x, y, lik, batch_shape=torch.Size()):
            super().__init__(x, y, lik)
            self.mean_module = g.means.ConstantMean(batch_shape=batch_shape, constant_prior=g.priors.SmoothedBoxPrior(-1, 1))
            self.covar_module = g.kernels.ScaleKernel(
                g.kernels.RBFKernel(batch_shape=batch_shape,
                                    lengthscale_prior=g.priors.NormalPrior(loc=torch.zeros(*batch_shape, 1, 1), scale=torch.ones(*batch_shape, 1, 1))),
                batch_shape=batch_shape, outputscale_prior=g.priors.SmoothedBoxPrior(-2, 2))

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    likelihood = g.likelihoods.GaussianLikelihood(noise_prior=g.priors.NormalPrior(loc=torch.zeros(2), scale=torch.ones(2)), batch_shape=torch.Size([2])).to(dev)
    gp_model = ExactGPModel(train_x12, train_y12, likelihood, batch_shape=torch.Size([2])).to(dev)
    mll = g.ExactMarginalLogLikelihood(likelihood, gp_model)
    gp_model.train()
    likelihood.train()
    optimizer = torch.optim.Adam(gp_model.parameters(), lr=0.1)
    with g.settings.max_cholesky_size(0), g.settings.debug(False):
        for _ in range(50):
            optimizer.zero_grad()
            output = gp_model(train_x12)
            loss = -mll(output, train_y12, train_x12).sum()
            loss.backward()
            optimizer.step()
            for param in gp_model.parameters():
                assert param.grad is not None and param.grad.norm().item() > 0
        gp_model.eval()
        likelihood.eval()
        preds1 = likelihood(gp_model(test_x1.to(dev))).mean           # non-batch test set against the batch model
        assert float(torch.mean(torch.abs(test_y1.to(dev) - preds1[0]))) < 0.1
        batch_predictions = likelihood(gp_model(test_x12))
        assert float(torch.mean(torch.abs(test_y1.to(dev) - batch_predictions.mean[0]))) < 0.1
        assert float(torch.mean(torch.abs(test_y2.to(dev) - batch_predictions.mean[1]))) < 0.1
        assert bool((batch_predictions.variance > 0).all())
        # derivatives with respect to the test inputs, batch and non-batch
        for tx in (test_x12, test_x1.to(dev)):
            test_x_param = torch.nn.Parameter(tx.clone())
            likelihood(gp_model(test_x_param)).mean.sum().backward()
            assert test_x_param.grad is not None and float(test_x_param.grad.abs().sum()) > 0


@pytest.mark.parametrize("kind,ard", [("rbf", False), ("matern52", True), ("matern12", False), ("rq", False)])
def test_small_members_are_evaluated_stacked(kind, ard, dev):
    """Members below ``max_cholesky_size``: the whole batch goes through ``batched.BatchedCholeskyInvQuadLogdetFn`` (one dense-generation
    launch + batched Cholesky + one derivative launch, ``csrc/extra_batch.hip``).  Value and EVERY hyper-parameter gradient
    (lengthscale(s), outputscale, noise, constant mean, RQ alpha) == the launch plan over members (``batched_small_members(False)``)
    == dense float64 autograd per member."""
    import gpytorch_amd as g
    from gpytorch_amd import batched

    b, n, d = 5, 140, 3
    gen = torch.Generator().manual_seed(3)
    X = torch.rand(b, n, d, generator=gen)
    Y = torch.sin(3 * X.sum(-1)) + 0.1 * torch.randn(b, n, generator=gen)
    bs = torch.Size([b])

    def base():
        if kind == "rbf":
            return g.kernels.RBFKernel(batch_shape=bs, ard_num_dims=d if ard else None)
        if kind == "rq":
            return g.kernels.RQKernel(batch_shape=bs, ard_num_dims=d if ard else None)
        return g.kernels.MaternKernel(nu=OK.KINDS[kind], batch_shape=bs, ard_num_dims=d if ard else None)

    class M(g.models.ExactGP):
        def __init__(self, x, y, lik):
            super().__init__(x, y, lik)
            self.mean_module = g.means.ConstantMean(batch_shape=bs)
            self.covar_module = g.kernels.ScaleKernel(base(), batch_shape=bs)

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    nls = d if ard else 1
    ls = 0.3 + 0.5 * torch.rand(b, 1, nls, generator=gen)
    os_ = 0.7 + torch.rand(b, generator=gen)
    nz = 0.05 + 0.2 * torch.rand(b, 1, generator=gen)
    al = 0.8 + 2.0 * torch.rand(b, 1, generator=gen)
    mu = 0.3 * torch.randn(b, generator=gen)

    def evaluate(stacked):
        lik = g.likelihoods.GaussianLikelihood(batch_shape=bs).to(dev)
        m = M(X.to(dev), Y.to(dev), lik).to(dev)
        m.covar_module.base_kernel.lengthscale = ls
        m.covar_module.outputscale = os_
        lik.noise = nz
        m.mean_module.constant = mu
        if kind == "rq":
            m.covar_module.base_kernel.alpha = al
        mll = g.ExactMarginalLogLikelihood(lik, m)
        m.train()
        lik.train()
        calls = []
        orig = batched.BatchedCholeskyInvQuadLogdetFn.apply
        batched.BatchedCholeskyInvQuadLogdetFn.apply = lambda *a: (calls.append(1), orig(*a))[1]
        try:
            with g.settings.batched_small_members(stacked):
                val = mll(m(m.train_inputs[0]), m.train_targets)
                val.sum().backward()
        finally:
            batched.BatchedCholeskyInvQuadLogdetFn.apply = orig
        assert len(calls) == (1 if stacked else 0)
        k = m.covar_module.base_kernel
        grads = [k.raw_lengthscale.grad, m.covar_module.raw_outputscale.grad, lik.noise_covar.raw_noise.grad, m.mean_module.raw_constant.grad]
        if kind == "rq":
            grads.append(k.raw_alpha.grad)
        return val.detach().double().cpu(), [x.detach().double().cpu().reshape(b, -1) for x in grads]

    v1, g1 = evaluate(True)
    v0, g0 = evaluate(False)
    assert v1.shape == torch.Size([b])
    assert torch.allclose(v1, v0, rtol=2e-5, atol=2e-5)
    for a, c in zip(g1, g0):
        assert torch.allclose(a, c, rtol=3e-3, atol=3e-3 * float(c.abs().max())), (kind, a, c)
    # dense float64 autograd per member (softplus chain rule on the raw parameters: d softplus(raw) = 1 - exp(-value))
    for i in range(b):
        p = [ls[i].double().clone().requires_grad_(True), os_[i].double().clone().requires_grad_(True), nz[i].double().reshape(()).clone().requires_grad_(True),
             mu[i].double().clone().requires_grad_(True), al[i].double().reshape(()).clone().requires_grad_(True)]
        Xi, Yi = X[i].double(), Y[i].double()
        if kind == "rq":
            Kd = OK.rq(Xi, Xi, p[0], p[4], x1_eq_x2=True, direct=True)
        else:
            Kd = OK.kernel_matrix(kind, Xi - Xi.mean(0), Xi - Xi.mean(0), p[0], 1.0, x1_eq_x2=True, direct=True)
        Kh = p[1] * Kd + p[2] * torch.eye(n, dtype=torch.float64)
        ref = OG.dense_log_prob(Kh, Yi - p[3]) / n
        gr = torch.autograd.grad(ref, p[:5] if kind == "rq" else p[:4], allow_unused=True)
        assert abs(float(v1[i]) - float(ref)) < 2e-4 * max(1.0, abs(float(ref))), (kind, i, float(v1[i]), float(ref))
        chain = [1.0 - torch.exp(-ls[i].double()).reshape(-1), 1.0 - math.exp(-float(os_[i])), 1.0 - math.exp(-(float(nz[i]) - 1e-4)), 1.0,
                 1.0 - math.exp(-float(al[i]))]
        for q, (gg, rr) in enumerate(zip(g1, gr)):
            want = (rr.reshape(-1) * chain[q]).reshape(-1)
            assert torch.allclose(gg[i].reshape(-1), want, rtol=5e-3, atol=5e-3 * float(want.abs().max()) + 1e-7), (kind, i, q, gg[i], want)


def test_batch_fixed_noise_with_learned_second_noise_stacked(dev):
    """A batch of FixedNoiseGaussianLikelihood GPs (``gaussian_likelihood.py:245-362``; fixed noise [b, n] + ``learn_additional_noise``): the
    fixed vector rides on every member's diagonal, the learned scalar keeps its gradient -- stacked evaluation == launch plan over members
    == dense float64 per member."""
    import gpytorch_amd as g

    b, n, d = 3, 120, 2
    gen = torch.Generator().manual_seed(9)
    X = torch.rand(b, n, d, generator=gen)
    Y = torch.sin(4 * X[..., 0]) + 0.1 * torch.randn(b, n, generator=gen)
    fixed = 0.02 + 0.1 * torch.rand(b, n, generator=gen)
    bs = torch.Size([b])

    class M(g.models.ExactGP):
        def __init__(self, x, y, lik):
            super().__init__(x, y, lik)
            self.mean_module = g.means.ZeroMean()
            self.covar_module = g.kernels.ScaleKernel(g.kernels.RBFKernel(batch_shape=bs), batch_shape=bs)

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    ls = torch.tensor([0.3, 0.45, 0.6]).view(b, 1, 1)
    os_ = torch.tensor([1.1, 0.8, 1.4])

    def evaluate(stacked):
        lik = g.likelihoods.FixedNoiseGaussianLikelihood(fixed.to(dev), learn_additional_noise=True).to(dev)
        m = M(X.to(dev), Y.to(dev), lik).to(dev)
        m.covar_module.base_kernel.lengthscale = ls
        m.covar_module.outputscale = os_
        lik.second_noise = 0.07
        mll = g.ExactMarginalLogLikelihood(lik, m)
        m.train()
        lik.train()
        with g.settings.batched_small_members(stacked):
            val = mll(m(m.train_inputs[0]), m.train_targets)
            val.sum().backward()
        return (val.detach().double().cpu(), m.covar_module.base_kernel.raw_lengthscale.grad.double().cpu().reshape(-1),
                lik.second_noise_covar.raw_noise.grad.double().cpu().reshape(-1))

    v1, gl1, gn1 = evaluate(True)
    v0, gl0, gn0 = evaluate(False)
    assert torch.allclose(v1, v0, rtol=2e-5, atol=2e-5)
    assert torch.allclose(gl1, gl0, rtol=3e-3, atol=1e-6)
    assert float(gn1.abs().sum()) > 0 and torch.allclose(gn1, gn0, rtol=3e-3, atol=1e-6)
    for i in range(b):
        Kh = float(os_[i]) * OK.kernel_matrix("rbf", X[i].double(), X[i].double(), float(ls[i]), 1.0, x1_eq_x2=True, direct=True) \
            + torch.diag(fixed[i].double() + 0.07)
        ref = OG.dense_log_prob(Kh, Y[i].double()) / n
        assert abs(float(v1[i]) - float(ref)) < 2e-4 * max(1.0, abs(float(ref))), (i, float(v1[i]), float(ref))


def test_mid_size_members_take_the_stacked_dense_path(dev):
    """Round 4 (``settings.batched_small_members.max_size``): members above ``max_cholesky_size`` (800) but below 3000 points are still evaluated
    stacked -- dense generation, batched float64 Cholesky, one derivative launch -- instead of one BBMM evaluation per member (measured 4-9 x
    faster for 64 members of 1000-2000 points, and exact).  Value and lengthscale / outputscale / noise gradients == dense float64 autograd per
    member; ``max_size = 0`` restores the member loop (stochastic log-det: compared at its own accuracy)."""
    import gpytorch_amd as g
    from gpytorch_amd import batched

    b, n, d = 3, 1000, 3
    gen = torch.Generator().manual_seed(11)
    X = torch.rand(b, n, d, generator=gen)
    Y = torch.sin(3 * X.sum(-1)) + 0.1 * torch.randn(b, n, generator=gen)
    bs = torch.Size([b])
    ls = 0.3 + 0.3 * torch.rand(b, 1, 1, generator=gen)
    os_ = 0.8 + 0.5 * torch.rand(b, generator=gen)
    nz = 0.05 + 0.1 * torch.rand(b, 1, generator=gen)

    class M(g.models.ExactGP):
        def __init__(self, x, y, lik):
            super().__init__(x, y, lik)
            self.mean_module = g.means.ConstantMean(batch_shape=bs)      # (constant initialised to 0)
            self.covar_module = g.kernels.ScaleKernel(g.kernels.RBFKernel(batch_shape=bs), batch_shape=bs)

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    def evaluate(max_size):
        lik = g.likelihoods.GaussianLikelihood(batch_shape=bs).to(dev)
        m = M(X.to(dev), Y.to(dev), lik).to(dev)
        m.covar_module.base_kernel.lengthscale = ls
        m.covar_module.outputscale = os_
        lik.noise = nz
        mll = g.ExactMarginalLogLikelihood(lik, m)
        m.train()
        lik.train()
        calls = []
        orig = batched.BatchedCholeskyInvQuadLogdetFn.apply
        batched.BatchedCholeskyInvQuadLogdetFn.apply = lambda *a: (calls.append(1), orig(*a))[1]
        old = g.settings.batched_small_members.max_size
        g.settings.batched_small_members.max_size = max_size
        try:
            torch.manual_seed(0)
            with g.settings.num_trace_samples(64), g.settings.cg_tolerance(1e-3):
                val = mll(m(m.train_inputs[0]), m.train_targets)
                val.sum().backward()
        finally:
            batched.BatchedCholeskyInvQuadLogdetFn.apply = orig
            g.settings.batched_small_members.max_size = old
        k = m.covar_module.base_kernel
        grads = [k.raw_lengthscale.grad, m.covar_module.raw_outputscale.grad, lik.noise_covar.raw_noise.grad]
        return len(calls), val.detach().double().cpu(), [x.detach().double().cpu().reshape(b, -1) for x in grads]

    c1, v1, g1 = evaluate(3000)
    c0, v0, g0 = evaluate(0)
    assert c1 == 1 and c0 == 0
    assert torch.allclose(v1, v0, rtol=0, atol=2e-2)          # the member loop's 64-probe log-det estimate
    for i in range(b):
        p = [ls[i].double().clone().requires_grad_(True), os_[i].double().clone().requires_grad_(True), nz[i].double().reshape(()).clone().requires_grad_(True)]
        Xi, Yi = X[i].double(), Y[i].double()
        Kh = p[1] * OK.kernel_matrix("rbf", Xi - Xi.mean(0), Xi - Xi.mean(0), p[0], 1.0, x1_eq_x2=True, direct=True) + p[2] * torch.eye(n, dtype=torch.float64)
        ref = OG.dense_log_prob(Kh, Yi) / n
        gr = torch.autograd.grad(ref, p)
        assert abs(float(v1[i]) - float(ref)) < 2e-4 * max(1.0, abs(float(ref))), (i, float(v1[i]), float(ref))
        chain = [1.0 - torch.exp(-ls[i].double()).reshape(-1), 1.0 - math.exp(-float(os_[i])), 1.0 - math.exp(-(float(nz[i]) - 1e-4))]
        for q, (gg, rr) in enumerate(zip(g1, gr)):
            want = (rr.reshape(-1) * chain[q]).reshape(-1)
            assert torch.allclose(gg[i].reshape(-1), want, rtol=5e-3, atol=5e-3 * float(want.abs().max()) + 1e-7), (i, q, gg[i], want)
