"""``test/examples/test_simple_gp_regression.py:47-330`` -- the reference's basic exact-GP behaviours -- restated against this repository's API (same data,
same hyper-parameter settings, same assertions).  Shared by the CPU wiring test (native entry points doubled) and the device test."""
from math import exp, pi

import torch


def _model_class(g):
    class ExactGPModel(g.models.ExactGP):
        def __init__(self, train_inputs, train_targets, likelihood):
            super().__init__(train_inputs, train_targets, likelihood)
            self.mean_module = g.means.ConstantMean()
            self.covar_module = g.kernels.ScaleKernel(g.kernels.RBFKernel())

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    return ExactGPModel


def _data(dev, num_data=11):
    train_x = torch.linspace(0, 1, num_data, device=dev)
    test_x = torch.linspace(0, 1, 51, device=dev)
    return train_x, test_x, torch.sin(train_x * (2 * pi)), torch.sin(test_x * (2 * pi))


def _close(a, b, atol=1e-5):
    torch.testing.assert_close(a, b, rtol=1e-4, atol=atol)


def case_prior(g, dev):                                                             # :47-78
    from gpytorch_amd.module import Positive

    train_x, _, _, _ = _data(dev)
    SmoothedBoxPrior = g.priors.SmoothedBoxPrior
    likelihood = g.likelihoods.GaussianLikelihood(noise_prior=SmoothedBoxPrior(exp(-3), exp(3), sigma=0.1), noise_constraint=Positive())
    model = _model_class(g)(None, None, likelihood)
    model.covar_module.base_kernel.register_prior("lengthscale_prior", SmoothedBoxPrior(exp(-10), exp(10), sigma=0.5), "raw_lengthscale")
    model.mean_module.initialize(constant=1.5)
    model.covar_module.base_kernel.initialize(lengthscale=1)
    likelihood.initialize(noise=0)
    model.to(dev)
    likelihood.to(dev)
    model.eval()
    likelihood.eval()
    pred = likelihood(model(train_x))                                               # no training data: the model predicts in prior mode
    correct_variance = model.covar_module.outputscale + likelihood.noise
    _close(pred.mean, torch.full_like(pred.mean, 1.5))
    _close(pred.variance, correct_variance.squeeze().expand_as(pred.variance))


def case_recursive_initialize(g, dev):                                              # :85-102
    train_x, _, train_y, _ = _data(dev)
    M = _model_class(g)
    m1, m2 = M(train_x, train_y, g.likelihoods.GaussianLikelihood()), M(train_x, train_y, g.likelihoods.GaussianLikelihood())
    m1.initialize(**{"likelihood.noise": 1e-2, "covar_module.base_kernel.lengthscale": 1e-1})
    m2.likelihood.initialize(noise=1e-2)
    m2.covar_module.base_kernel.initialize(lengthscale=1e-1)
    assert torch.equal(m1.likelihood.noise, m2.likelihood.noise)
    assert torch.equal(m1.covar_module.base_kernel.lengthscale, m2.covar_module.base_kernel.lengthscale)


def case_posterior_without_optimization(g, dev):                                    # :104-136
    from gpytorch_amd.module import Positive

    train_x, test_x, train_y, _ = _data(dev)
    likelihood = g.likelihoods.GaussianLikelihood(noise_constraint=Positive())     # (this case wants a noise < 1e-4)
    model = _model_class(g)(train_x, train_y, likelihood)
    model.covar_module.base_kernel.initialize(lengthscale=exp(-15))
    likelihood.initialize(noise=exp(-15))
    model.to(dev)
    likelihood.to(dev)
    model.eval()
    likelihood.eval()
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with g.settings.debug(False), torch.no_grad():
            pred = likelihood(model(train_x))                                       # the posterior fits all the data ...
            _close(pred.mean, train_y)
            _close(pred.variance, torch.zeros_like(pred.variance))
            out = model(torch.tensor([1.1], device=dev))                            # ... and nothing else
            _close(out.mean, torch.zeros_like(out.mean))
            _close(out.variance, model.covar_module.outputscale.expand_as(out.variance))


def case_skip_variances(g, dev):                                                    # :143-187
    train_x, test_x, train_y, _ = _data(dev)
    likelihood = g.likelihoods.GaussianLikelihood()
    model = _model_class(g)(train_x, train_y, likelihood).to(dev)
    likelihood.to(dev)
    model.eval()
    likelihood.eval()
    for fast in (True, False):
        with torch.no_grad(), g.settings.fast_pred_var(fast):
            with g.settings.skip_posterior_variances(True):
                mean_skip = model(test_x).mean
            assert torch.allclose(mean_skip, model(test_x).mean)
            assert torch.allclose(mean_skip, likelihood(model(test_x)).mean)


def case_single_training_point(g, dev):                                             # :189-215
    train_x, test_x, train_y, _ = _data(dev)
    likelihood = g.likelihoods.GaussianLikelihood()
    model = _model_class(g)(train_x[0].unsqueeze(-1).unsqueeze(-1), train_y[0].unsqueeze(-1), likelihood).to(dev)
    model.eval()
    likelihood.eval()
    with torch.no_grad():
        with g.settings.fast_pred_var():
            p = model(test_x)
            assert not torch.isnan(p.mean).any() and not torch.isnan(p.variance).any()
        model.train()
        model.eval()
        p = model(test_x)
        assert not torch.isnan(p.mean).any() and not torch.isnan(p.variance).any()


def _trained(g, dev, prior=False):
    train_x, test_x, train_y, test_y = _data(dev)
    likelihood = g.likelihoods.GaussianLikelihood(noise_prior=g.priors.SmoothedBoxPrior(exp(-3), exp(3), sigma=0.1)) if prior else g.likelihoods.GaussianLikelihood()
    model = _model_class(g)(train_x, train_y, likelihood)
    mll = g.ExactMarginalLogLikelihood(likelihood, model)
    model.covar_module.base_kernel.initialize(lengthscale=exp(1))
    model.mean_module.initialize(constant=0)
    likelihood.initialize(noise=exp(1))
    model.to(dev)
    likelihood.to(dev)
    model.train()
    likelihood.train()
    optimizer = torch.optim.Adam(model.parameters(), lr=0.15)
    for _ in range(50):
        optimizer.zero_grad()
        with g.settings.debug(False):
            loss = -mll(model(train_x), train_y)
        loss.backward()
        optimizer.step()
    for param in model.parameters():
        assert param.grad is not None and param.grad.norm().item() > 0
    optimizer.step()
    return model, likelihood, train_x, test_x, train_y, test_y


def case_posterior_with_optimization(g, dev):                                       # :217-257
    with g.settings.fast_pred_var():
        model, likelihood, _, test_x, _, test_y = _trained(g, dev, prior=True)
    model.eval()
    likelihood.eval()
    with torch.no_grad(), g.settings.skip_posterior_variances(True):
        pred = likelihood(model(test_x))
    assert torch.mean(torch.abs(test_y - pred.mean)).item() < 0.05


def case_fantasy_updates(g, dev):                                                   # :264-323
    model, likelihood, train_x, test_x, train_y, _ = _trained(g, dev)
    train_x.requires_grad = True
    model.set_train_data(train_x, train_y)
    with g.settings.fast_pred_var(), g.settings.detach_test_caches(False):
        model.eval()
        likelihood.eval()
        pred = likelihood(model(test_x))
        pred.mean.sum().backward()
        real_grad = train_x.grad[5:].clone()
        train_x.grad = None
        train_x.requires_grad = False
        model.set_train_data(train_x, train_y)
        model.set_train_data(train_x[:5], train_y[:5], strict=False)                # cut the data down, add it back through the fantasy interface
        likelihood(model(test_x))
        fantasy_x = train_x[5:].clone().detach().requires_grad_(True)
        fant_model = model.get_fantasy_model(fantasy_x, train_y[5:])
        fant_pred = likelihood(fant_model(test_x))
        _close(pred.mean, fant_pred.mean, atol=1e-4)
        fant_pred.mean.sum().backward()
        assert fantasy_x.grad is not None
        assert torch.norm(real_grad - fantasy_x.grad) / fantasy_x.grad.norm() < 15e-1


CASES = [case_prior, case_recursive_initialize, case_posterior_without_optimization, case_skip_variances, case_single_training_point,
         case_posterior_with_optimization, case_fantasy_updates]


def case_fixed_noise_fantasy_updates(g, dev):                                       # test/examples/test_fixed_noise_fanatasy_updates.py:51-113
    from gpytorch_amd.likelihoods import FixedGaussianNoise

    train_x, test_x, train_y, test_y = _data(dev)
    noise, test_noise = torch.full_like(train_y, 2e-4), torch.full_like(test_y, 3e-4)
    likelihood = g.likelihoods.FixedNoiseGaussianLikelihood(noise)
    model = _model_class(g)(train_x, train_y, likelihood)
    mll = g.ExactMarginalLogLikelihood(likelihood, model)
    model.covar_module.base_kernel.initialize(lengthscale=exp(1))
    model.mean_module.initialize(constant=0)
    model.to(dev)
    likelihood.to(dev)
    model.train()
    likelihood.train()
    optimizer = torch.optim.Adam(model.parameters(), lr=0.15)
    for _ in range(50):
        optimizer.zero_grad()
        with g.settings.debug(False):
            loss = -mll(model(train_x), train_y)
        loss.backward()
        optimizer.step()
    for param in model.parameters():
        assert param.grad is not None and param.grad.norm().item() > 0
    optimizer.step()
    train_x.requires_grad = True
    model.set_train_data(train_x, train_y)
    with g.settings.fast_pred_var(), g.settings.detach_test_caches(False):
        model.eval()
        likelihood.eval()
        pred = likelihood(model(test_x), noise=test_noise)
        pred.mean.sum().backward()
        real_grad = train_x.grad[5:].clone()
        train_x.grad = None
        train_x.requires_grad = False
        model.set_train_data(train_x, train_y)
        model.set_train_data(train_x[:5], train_y[:5], strict=False)
        model.likelihood.noise_covar = FixedGaussianNoise(noise=noise[:5])
        likelihood(model(test_x), noise=test_noise)
        fantasy_x = train_x[5:].clone().detach().requires_grad_(True)
        fant_model = model.get_fantasy_model(fantasy_x, train_y[5:], noise=noise[5:])
        fant_pred = likelihood(fant_model(test_x), noise=test_noise)
        _close(pred.mean, fant_pred.mean, atol=1e-4)
        fant_pred.mean.sum().backward()
        assert fantasy_x.grad is not None
        assert torch.norm(real_grad - fantasy_x.grad) / fantasy_x.grad.norm() < 15e-1


def _missing(g, dev, batch):                                                        # test/examples/test_missing_data.py:75-175 ("mask" policy)
    bs = torch.Size((2,)) if batch else torch.Size(())

    class SingleGPModel(g.models.ExactGP):
        def __init__(self, x, y, likelihood):
            super().__init__(x, y, likelihood)
            self.mean_module = g.means.ConstantMean(batch_shape=bs)
            self.covar_module = g.kernels.ScaleKernel(g.kernels.RBFKernel(batch_shape=bs))

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    if batch:
        train_x = torch.stack([torch.linspace(0, 1, 41), torch.linspace(1, 2, 41)]).reshape(2, 41, 1).to(dev)
        test_x = torch.stack([torch.linspace(0, 1, 51), torch.linspace(1, 2, 51)]).reshape(2, 51, 1).to(dev)
    else:
        train_x, test_x = torch.linspace(0, 1, 41, device=dev), torch.linspace(0, 1, 51, device=dev)
    train_y = torch.sin(2 * torch.pi * train_x).squeeze()
    train_y = train_y + torch.normal(0, 0.01, train_y.shape).to(dev)
    test_y = torch.sin(2 * torch.pi * test_x).squeeze()
    if batch:
        train_y[0, ::4] = torch.nan
    else:
        train_y[::4] = torch.nan
    likelihood = g.likelihoods.GaussianLikelihood(batch_shape=bs).to(dev)
    model = SingleGPModel(train_x, train_y, likelihood).to(dev)
    mll = g.ExactMarginalLogLikelihood(likelihood, model)
    optimizer = torch.optim.Adam(model.parameters(), lr=0.15)
    with g.settings.observation_nan_policy("mask"):
        model.train()
        likelihood.train()
        for _ in range(30):
            optimizer.zero_grad()
            output = model(train_x)
            loss = -mll(output, train_y).sum()
            assert not torch.isnan(output.mean).any() and not torch.isnan(loss)
            loss.backward()
            optimizer.step()
        model.eval()
        likelihood.eval()
        with torch.no_grad():
            prediction = model(test_x)
            assert not torch.isnan(prediction.mean).any() and not torch.isnan(prediction.covariance_matrix).any()
            torch.testing.assert_close(prediction.mean, test_y, rtol=1e-4, atol=0.2)
    import warnings

    with torch.no_grad(), g.settings.observation_nan_policy("fill"), warnings.catch_warnings(record=True) as caught:      # :114-131: the other policy, warned
        warnings.simplefilter("always")
        prediction = model(test_x)
        assert any(issubclass(w.category, RuntimeWarning) and "fill" in str(w.message) for w in caught)
        assert not torch.isnan(prediction.mean).any() and not torch.isnan(prediction.covariance_matrix).any()
        torch.testing.assert_close(prediction.mean, test_y, rtol=1e-4, atol=0.2)


def case_missing_data_single(g, dev):
    _missing(g, dev, False)


def case_missing_data_single_batch(g, dev):
    _missing(g, dev, True)


CASES += [case_fixed_noise_fantasy_updates, case_missing_data_single, case_missing_data_single_batch]


def _batch_data(dev):                                                               # test/examples/test_batch_gp_regression.py:20-35
    import math

    x1, x2 = torch.linspace(0, 2, 11).unsqueeze(-1), torch.linspace(0, 1, 11).unsqueeze(-1)
    y1 = torch.sin(x1 * (2 * math.pi)).squeeze()
    y2 = torch.sin(x2 * (2 * math.pi)).squeeze()
    y1, y2 = y1 + 0.01 * torch.randn_like(y1), y2 + 0.01 * torch.randn_like(y2)
    t1, t2 = torch.linspace(0, 2, 51).unsqueeze(-1), torch.linspace(0, 1, 51).unsqueeze(-1)
    ty1, ty2 = torch.sin(t1 * (2 * math.pi)).squeeze(), torch.sin(t2 * (2 * math.pi)).squeeze()
    mv = lambda *ts: [t.to(dev) for t in ts]  # noqa: E731
    return mv(x1, y1, t1, ty1, x2, y2, t2, ty2, torch.stack([x1, x2]), torch.stack([y1, y2]), torch.stack([t1, t2]))


def _batch_model_class(g):
    P = g.priors

    class ExactGPModel(g.models.ExactGP):                                           # :38-57 (priors included)
        def __init__(self, x, y, likelihood, batch_shape=torch.Size()):
            super().__init__(x, y, likelihood)
            self.mean_module = g.means.ConstantMean(batch_shape=batch_shape, constant_prior=P.SmoothedBoxPrior(-1, 1))
            self.covar_module = g.kernels.ScaleKernel(
                g.kernels.RBFKernel(batch_shape=batch_shape,
                                    lengthscale_prior=P.NormalPrior(loc=torch.zeros(*batch_shape, 1, 1), scale=torch.ones(*batch_shape, 1, 1))),
                batch_shape=batch_shape, outputscale_prior=P.SmoothedBoxPrior(-2, 2))

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    return ExactGPModel


def _batch_checks(g, model, likelihood, t1, ty1, ty2, t12, first_is_batch):
    model.eval()
    likelihood.eval()
    preds1 = likelihood(model(t1)).mean
    assert torch.mean(torch.abs(ty1 - (preds1[0] if first_is_batch else preds1))).item() < 0.1
    bp = likelihood(model(t12))
    assert torch.mean(torch.abs(ty1 - bp.mean[0])).item() < 0.1 and torch.mean(torch.abs(ty2 - bp.mean[1])).item() < 0.1
    for tx in (t1, t12):                                                            # derivatives with respect to the test inputs, both modes
        tp = torch.nn.Parameter(tx.detach().clone())
        likelihood(model(tp)).mean.sum().backward()
        assert tp.grad is not None


def case_train_on_single_set_test_on_batch(g, dev):                                 # :72-123
    x1, y1, t1, ty1, x2, y2, t2, ty2, x12, y12, t12 = _batch_data(dev)
    likelihood = g.likelihoods.GaussianLikelihood().to(dev)
    model = _batch_model_class(g)(x1, y1, likelihood).to(dev)
    mll = g.ExactMarginalLogLikelihood(likelihood, model)
    model.train()
    likelihood.train()
    optimizer = torch.optim.Adam(model.parameters(), lr=0.1)
    for _ in range(75):
        optimizer.zero_grad()
        loss = -mll(model(x1), y1).sum()
        loss.backward()
        optimizer.step()
        for param in model.parameters():
            assert param.grad is not None and param.grad.norm().item() > 0
    _batch_checks(g, model, likelihood, t1, ty1, ty2, t12, False)


def case_train_on_batch_shared_hypers_over_batch(g, dev):                           # :180-232
    x1, y1, t1, ty1, x2, y2, t2, ty2, x12, y12, t12 = _batch_data(dev)
    likelihood = g.likelihoods.GaussianLikelihood(noise_prior=g.priors.NormalPrior(loc=torch.zeros(2), scale=torch.ones(2))).to(dev)
    model = _batch_model_class(g)(x12, y12, likelihood).to(dev)
    mll = g.ExactMarginalLogLikelihood(likelihood, model)
    model.train()
    likelihood.train()
    optimizer = torch.optim.Adam(model.parameters(), lr=0.1)
    for _ in range(50):
        optimizer.zero_grad()
        loss = -mll(model(x12), y12, x12).sum()
        loss.backward()
        optimizer.step()
        for param in model.parameters():
            assert param.grad is not None and param.grad.norm().item() > 0
    _batch_checks(g, model, likelihood, t1, ty1, ty2, t12, True)


CASES += [case_train_on_single_set_test_on_batch, case_train_on_batch_shared_hypers_over_batch]


def _multitask_model(g, num_tasks, rank, scale=False):
    class MultitaskGPModel(g.models.ExactGP):
        def __init__(self, x, y, likelihood):
            super().__init__(x, y, likelihood)
            self.mean_module = g.means.MultitaskMean(g.means.ConstantMean(), num_tasks=num_tasks)
            base = g.kernels.ScaleKernel(g.kernels.RBFKernel()) if scale else g.kernels.RBFKernel()
            self.covar_module = g.kernels.MultitaskKernel(base, num_tasks=num_tasks, rank=rank)

        def forward(self, x):
            return g.distributions.MultitaskMultivariateNormal(self.mean_module(x), self.covar_module(x))

    return MultitaskGPModel


def case_kronecker_multitask_mean_abs_error(g, dev):                                # test/examples/test_kronecker_multitask_gp_regression.py:20-91
    train_x = torch.linspace(0, 1, 100, device=dev)
    train_y = torch.stack([torch.sin(train_x * (2 * pi)) + torch.randn(100, device=dev) * 0.1,
                           torch.cos(train_x * (2 * pi)) + torch.randn(100, device=dev) * 0.1], -1)
    likelihood = g.likelihoods.MultitaskGaussianLikelihood(num_tasks=2).to(dev)
    model = _multitask_model(g, 2, 2)(train_x, train_y, likelihood).to(dev)
    model.train()
    likelihood.train()
    optimizer = torch.optim.Adam(model.parameters(), lr=0.1)
    mll = g.mlls.ExactMarginalLogLikelihood(likelihood, model)
    for _ in range(50):
        optimizer.zero_grad()
        loss = -mll(model(train_x), train_y)
        loss.backward()
        optimizer.step()
    model.eval()
    likelihood.eval()
    test_x = torch.linspace(0, 1, 51, device=dev)
    with torch.no_grad():
        preds = likelihood(model(test_x)).mean
    assert torch.mean(torch.abs(torch.sin(test_x * (2 * pi)) - preds[:, 0])).item() < 0.05
    assert torch.mean(torch.abs(torch.cos(test_x * (2 * pi)) - preds[:, 1])).item() < 0.05


def case_missing_data_multitask(g, dev):                                            # test/examples/test_missing_data.py:177-195 ("mask" policy)
    num_tasks = 10
    train_x, test_x = torch.linspace(0, 1, 41, device=dev), torch.linspace(0, 1, 51, device=dev)
    coefficients = torch.rand(1, num_tasks, device=dev)
    train_y = torch.sin(2 * torch.pi * train_x)[:, None] * coefficients
    train_y = train_y + torch.normal(0, 0.01, train_y.shape).to(dev)
    test_y = torch.sin(2 * torch.pi * test_x)[:, None] * coefficients
    train_y[::3, : num_tasks // 2] = torch.nan
    train_y[::4, num_tasks // 2:] = torch.nan
    likelihood = g.likelihoods.MultitaskGaussianLikelihood(num_tasks).to(dev)
    model = _multitask_model(g, num_tasks, 1, scale=True)(train_x, train_y, likelihood).to(dev)
    mll = g.mlls.ExactMarginalLogLikelihood(likelihood, model)
    optimizer = torch.optim.Adam(model.parameters(), lr=0.15)
    with g.settings.observation_nan_policy("mask"):
        model.train()
        likelihood.train()
        for _ in range(30):
            optimizer.zero_grad()
            output = model(train_x)
            loss = -mll(output, train_y).sum()
            assert not torch.isnan(output.mean).any() and not torch.isnan(loss)
            loss.backward()
            optimizer.step()
        model.eval()
        likelihood.eval()
        with torch.no_grad():
            prediction = model(test_x)
            assert not torch.isnan(prediction.mean).any()
            torch.testing.assert_close(prediction.mean, test_y, rtol=1e-4, atol=0.2)


MULTITASK_CASES = [case_kronecker_multitask_mean_abs_error, case_missing_data_multitask]
