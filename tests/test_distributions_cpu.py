"""``test/distributions/test_multivariate_normal.py:20-370`` restated over ``gpytorch_amd.distributions.MultivariateNormal`` (dense and operator
covariances, the known answers 4.3157 / -4.8157, arithmetic, batches, indexing, base samples through a non-square root, expand / unsqueeze) and
``test/mlls/test_exact_marginal_log_likelihood.py:52-90`` (a batch of identical models evaluates to identical MLLs; the MLL is the marginal's log-probability
plus every prior's, divided by the number of data).  Host logic only; the MLL cases run on the CPU double of the native kernel evaluation."""
import pytest
import torch

import gpytorch_amd as g
from gpytorch_amd.operators import DenseLinearOperator, LinearOperator, RootLinearOperator

MVN = g.distributions.MultivariateNormal
ac = lambda a, b: torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)  # noqa: E731


@pytest.mark.parametrize("lazy", [False, True])
@pytest.mark.parametrize("dt", [torch.float, torch.double])
def test_multivariate_normal(lazy, dt):
    mean = torch.tensor([0, 1, 2], dtype=dt)
    cov = torch.diag(torch.tensor([1, 0.75, 1.5], dtype=dt))
    mvn = MVN(mean, DenseLinearOperator(cov) if lazy else cov)
    assert torch.is_tensor(mvn.covariance_matrix) and isinstance(mvn.lazy_covariance_matrix, LinearOperator) and mvn.islazy == lazy
    ac(mvn.variance, torch.diag(cov))
    ac(mvn.scale_tril, cov.sqrt())
    ac(mvn._unbroadcasted_scale_tril, torch.linalg.cholesky(cov))
    for new, m, c in ((mvn + 1, mvn.mean + 1, cov), (mvn * 2, mvn.mean * 2, cov * 4), (mvn / 2, mvn.mean / 2, cov / 4)):
        ac(new.mean, m)
        ac(new.covariance_matrix, c)
    assert abs(mvn.entropy().item() - 4.3157) < 1e-4
    assert abs(mvn.log_prob(torch.zeros(3, dtype=dt)).item() + 4.8157) < 1e-4
    ac(mvn.log_prob(torch.zeros(2, 3, dtype=dt)), torch.tensor([-4.8157, -4.8157], dtype=dt))
    lo, up = mvn.confidence_region()
    ac(lo, mvn.mean - 2 * mvn.stddev)
    ac(up, mvn.mean + 2 * mvn.stddev)
    assert mvn.sample().shape == (3,) and mvn.sample(torch.Size([2])).shape == (2, 3) and mvn.sample(torch.Size([2, 4])).shape == (2, 4, 3)
    with pytest.raises(RuntimeError, match="scalars"):
        mvn * torch.ones(3)


@pytest.mark.parametrize("dt", [torch.float, torch.double])
def test_multivariate_normal_batch(dt):
    mean = torch.tensor([0, 1, 2], dtype=dt).repeat(2, 1)
    cov = torch.diag(torch.tensor([1, 0.75, 1.5], dtype=dt)).repeat(2, 1, 1)
    mvn = MVN(mean, cov)
    ac(mvn.variance, torch.diagonal(cov, dim1=-2, dim2=-1))
    ac(mvn.entropy(), 4.3157 * torch.ones(2, dtype=dt))
    ac(mvn.log_prob(torch.zeros(2, 3, dtype=dt)), -4.8157 * torch.ones(2, dtype=dt))
    ac(mvn.log_prob(torch.zeros(2, 2, 3, dtype=dt)), -4.8157 * torch.ones(2, 2, dtype=dt))
    assert mvn.sample(torch.Size([2, 4])).shape == (2, 4, 2, 3)


def test_getitem():
    torch.manual_seed(0)
    shape = (2, 4, 3, 2)
    cov = torch.randn(*shape, shape[1])
    cov = cov @ cov.transpose(-1, -2)
    dist = MVN(torch.randn(*shape), cov)
    dc = dist.covariance_matrix
    for idx, want in (
        ((1,), dc[1]),
        ((Ellipsis, 1), dc[..., 1, 1].unsqueeze(-1) * torch.eye(shape[-2])),
        ((slice(None), [2, 3], slice(None), slice(1, None)), dc[:, [2, 3], :, 1:, 1:]),
        ((slice(None), slice(None), Ellipsis, [0, 1, 1, 0]), dc[..., [0, 1, 1, 0], :][..., [0, 1, 1, 0]]),
        ((1, 2, 2, Ellipsis), dc[1, 2, 2, :, :]),
        ((0, 1, Ellipsis, 2, 1), dc[0, 1, 2, 1, 1]),
    ):
        d = dist[idx[0] if len(idx) == 1 else idx]
        assert torch.equal(d.mean, dist.mean[tuple(i for i in idx if i is not Ellipsis) if len(idx) > 4 else idx])
        ac(d.covariance_matrix, want)


def test_base_sample_shape():
    a = torch.randn(5, 10)
    dist = MVN(torch.zeros(5), RootLinearOperator(a))
    assert dist.rsample(torch.Size((16,)), base_samples=torch.randn(16, 10)).shape == (16, 5)     # base samples of the ROOT's width
    with pytest.raises(RuntimeError):
        dist.rsample(torch.Size((16,)), base_samples=torch.randn(16, 5))
    dist = MVN(torch.zeros(5), DenseLinearOperator(a @ a.t()))
    assert dist.rsample(torch.Size((16,)), base_samples=torch.randn(16, 5)).shape == (16, 5)


@pytest.mark.parametrize("lazy", [False, True])
def test_expand_and_unsqueeze(lazy):
    mean, cov = torch.tensor([0.0, 1, 2]), torch.diag(torch.tensor([1, 0.75, 1.5]))
    mvn = MVN(mean, DenseLinearOperator(cov) if lazy else cov)
    mvn.scale_tril
    e = mvn.expand(torch.Size([2]))
    assert isinstance(e, MVN) and e.islazy == lazy and e.batch_shape == torch.Size([2]) and e.event_shape == mvn.event_shape
    assert torch.equal(e.mean, mean.expand(2, -1)) and torch.allclose(e.covariance_matrix, cov.expand(2, -1, -1))
    assert torch.allclose(e.scale_tril, mvn.scale_tril.expand(2, -1, -1)) and e.scale_tril.shape == (2, 3, 3)
    bs = torch.Size([2, 3])
    mvn = MVN(mean.expand(*bs, -1), DenseLinearOperator(cov.expand(*bs, -1, -1)) if lazy else cov.expand(*bs, -1, -1))
    for dim, expected in ((1, torch.Size([2, 1, 3])), (-1, torch.Size([2, 3, 1]))):
        new = mvn.unsqueeze(dim)
        assert isinstance(new, MVN) and new.islazy == lazy and new.batch_shape == expected
        assert new.covariance_matrix.shape == (*expected, 3, 3)
    with pytest.raises(IndexError):
        mvn.unsqueeze(3)


def _mll_model():
    from gpytorch_amd.constraints import GreaterThan

    GammaPrior = g.priors.GammaPrior

    class ExactGPModel(g.models.ExactGP):                      # test_exact_marginal_log_likelihood.py:17-49
        def __init__(self, x, y):
            bs = x.shape[:-2]
            noise_prior = GammaPrior(1.1, 0.05)
            mode = (noise_prior.concentration - 1) / noise_prior.rate
            likelihood = g.likelihoods.GaussianLikelihood(noise_prior=noise_prior, batch_shape=bs,
                                                          noise_constraint=GreaterThan(1e-4, transform=None, initial_value=mode))
            super().__init__(x, y, likelihood)
            self.mean_module = g.means.ConstantMean(batch_shape=bs)
            self.covar_module = g.kernels.ScaleKernel(
                g.kernels.MaternKernel(nu=2.5, ard_num_dims=x.shape[-1], batch_shape=bs, lengthscale_prior=GammaPrior(3.0, 6.0)),
                batch_shape=bs, outputscale_prior=GammaPrior(2.0, 0.15))

        def forward(self, x):
            return MVN(self.mean_module(x), self.covar_module(x))

    return ExactGPModel


def test_mll_batched_eval(monkeypatch):
    from tests.shim import cpu_backend

    cpu_backend.install(monkeypatch)
    M = _mll_model()
    x, y = torch.rand(10, 2), torch.randn(10)
    m = M(x, y)
    single = g.ExactMarginalLogLikelihood(m.likelihood, m)(m(x), y)
    xb, yb = x.expand(10, -1, -1), y.expand(10, -1)
    mb = M(xb, yb)
    batch = g.ExactMarginalLogLikelihood(mb.likelihood, mb)(mb(xb), yb)
    assert single.shape == torch.Size() and batch.shape == torch.Size([10])
    assert torch.allclose(single.expand(10), batch)


def test_mll_computation(monkeypatch):
    from tests.shim import cpu_backend

    cpu_backend.install(monkeypatch)
    x, y = torch.rand(10, 2), torch.rand(10)
    m = _mll_model()(x, y)
    out = m(x)
    value = g.ExactMarginalLogLikelihood(m.likelihood, m)(out, y)
    noise_prior = next(m.likelihood.named_priors())[2]
    outputscale_prior = next(m.covar_module.named_priors())[2]
    lengthscale_prior = next(m.covar_module.base_kernel.named_priors())[2]
    by_hand = sum([m.likelihood(out).log_prob(y), noise_prior.log_prob(m.likelihood.noise), outputscale_prior.log_prob(m.covar_module.outputscale),
                   lengthscale_prior.log_prob(m.covar_module.base_kernel.lengthscale).sum()]) / y.shape[0]
    assert torch.allclose(value, by_hand)
