"""``test/means/test_constant_mean.py:24-62`` + ``gpytorch/test/base_mean_test_case.py:19-46`` (shapes of a constant mean with and without a batch shape,
its prior and constraint), ``test/kernels/test_scale_kernel.py:129-141`` / ``test_periodic_kernel.py:91-103`` / ``test_index_kernel.py:9-21`` /
``test/likelihoods/test_gaussian_likelihood.py:23-27`` (a prior must be a Prior; modules with priors survive pickling), and a state-dict / pickle
round trip of a whole exact GP.  Host logic only."""
import io
import pickle

import pytest
import torch

import gpytorch_amd as g

K, P = g.kernels, g.priors


@pytest.mark.parametrize("bs", [None, torch.Size([3]), torch.Size([2, 3])])
def test_constant_mean(bs):
    def make(prior=None, constraint=None):
        return g.means.ConstantMean(constant_prior=prior, constant_constraint=constraint, batch_shape=bs or torch.Size([]))

    m = make()
    expect = (lambda *lead: torch.Size([*lead, 4])) if bs is None else (lambda *lead: torch.Size([*bs, 4]))
    assert m(torch.randn(4)).shape == expect() and m(torch.randn(4, 3)).shape == expect()
    if bs is None or len(bs) == 1:
        assert m(torch.randn(3, 4, 3)).shape == expect(3)
    assert m(torch.randn(2, 3, 4, 3)).shape == torch.Size([2, 3, 4])
    prior = P.NormalPrior(0.0, 1.0) if bs is None else P.NormalPrior(torch.zeros(bs), torch.ones(bs))
    m = make(prior=prior)
    assert m.mean_prior is prior
    pickle.loads(pickle.dumps(m))
    value = prior.sample()
    m._constant_closure(m, value)
    assert torch.equal(m.constant.data, value.reshape(m.constant.data.shape))
    assert torch.allclose(make().constant, torch.zeros(make().constant.shape))
    m = make(constraint=g.constraints.GreaterThan(1.5))
    assert torch.all(m.constant >= 1.5)
    m.constant = torch.full(bs or torch.Size([]), 1.65)
    assert torch.allclose(m.constant, torch.tensor(1.65).expand(m.constant.shape))


def test_zero_and_multitask_means():
    z = g.means.ZeroMean()
    assert z(torch.randn(4)).shape == (4,) and z(torch.randn(3, 4, 3)).shape == (3, 4) and float(z(torch.randn(4, 3)).abs().max()) == 0
    m = g.means.MultitaskMean(g.means.ConstantMean(), num_tasks=3)
    assert m(torch.randn(5, 2)).shape == (5, 3) and m(torch.randn(4, 5, 2)).shape == (4, 5, 3)
    m = g.means.MultitaskMean([g.means.ConstantMean(), g.means.ZeroMean(), g.means.ZeroMean()], num_tasks=3)
    assert m(torch.randn(5, 2)).shape == (5, 3)


@pytest.mark.parametrize("make", [
    lambda pr: K.ScaleKernel(K.RBFKernel(), outputscale_prior=pr),
    lambda pr: K.PeriodicKernel(period_length_prior=pr),
    lambda pr: K.RBFKernel(lengthscale_prior=pr),
    lambda pr: K.IndexKernel(num_tasks=1, prior=pr),
    lambda pr: g.likelihoods.GaussianLikelihood(noise_prior=pr),
], ids=["outputscale", "period_length", "lengthscale", "index_kernel", "noise"])
def test_prior_type_and_pickle(make):
    make(None)
    module = make(P.NormalPrior(0, 1))
    assert len(list(module.named_priors())) == 1
    pickle.loads(pickle.dumps(module))
    with pytest.raises(TypeError, match="Expected gpytorch.priors.Prior"):
        make(1)


class _Model(g.models.ExactGP):
    def __init__(self, x, y, likelihood):
        super().__init__(x, y, likelihood)
        self.mean_module = g.means.ConstantMean(constant_prior=P.NormalPrior(0, 1))
        self.covar_module = K.ScaleKernel(K.MaternKernel(nu=2.5, lengthscale_prior=P.GammaPrior(3.0, 6.0)), outputscale_prior=P.GammaPrior(2.0, 0.15))

    def forward(self, x):
        return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))


def test_model_state_dict_and_pickle_round_trip():
    x, y = torch.rand(10, 2), torch.randn(10)
    model = _Model(x, y, g.likelihoods.GaussianLikelihood(noise_prior=P.GammaPrior(1.1, 0.05)))
    model.covar_module.base_kernel.lengthscale = 0.37
    model.likelihood.noise = 0.2
    buf = io.BytesIO()
    torch.save(model.state_dict(), buf)
    buf.seek(0)
    fresh = _Model(x, y, g.likelihoods.GaussianLikelihood(noise_prior=P.GammaPrior(1.1, 0.05)))
    fresh.load_state_dict(torch.load(buf))
    assert torch.equal(fresh.covar_module.base_kernel.lengthscale, model.covar_module.base_kernel.lengthscale)
    assert torch.equal(fresh.likelihood.noise, model.likelihood.noise)
    clone = pickle.loads(pickle.dumps(model))
    assert torch.equal(clone.covar_module.base_kernel.lengthscale, model.covar_module.base_kernel.lengthscale)
    assert len(list(clone.named_priors())) == 4
    assert isinstance(clone.constraint_for_parameter_name("likelihood.noise_covar.raw_noise"), g.constraints.GreaterThan)
