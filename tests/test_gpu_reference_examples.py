"""The reference's OWN end-to-end checks, run 1:1 on the GPU path with CG forced (``max_cholesky_size(0)``: every
solve / log-det goes through the fused K*V + mBCG + SLQ + Lanczos kernels, however small n is):

  * white-noise regression, 75 Adam steps, MAE < 0.05          test/examples/test_white_noise_regression.py:56-102
  * ``fast_pred_var`` variance within 5 % of the noise         test/examples/test_simple_gp_regression.py:396-442
  * CG solve vs dense inverse, rtol 0.02 / atol 1e-5, hyper-   test/lazy/test_lazy_evaluated_kernel_tensor.py:69-113
    parameter gradients rtol 1e-3, rhs gradients rtol 0.03
  * batch-independent MLL == (log_prob + priors) / n            test/mlls/test_exact_marginal_log_likelihood.py:71-91
    (covered in test_gpu_extra.py::test_mll_with_priors_and_lbfgs_training)

Same data, same initialisation, same optimiser settings, same thresholds as the reference tests.
"""
import math
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu


def _sine_data(dev):
    train_x = torch.linspace(0, 1, 11, device=dev)
    train_y = torch.sin(train_x * (2 * math.pi))
    test_x = torch.linspace(0, 1, 51, device=dev)
    test_y = torch.sin(test_x * (2 * math.pi))
    return train_x, test_x, train_y, test_y


def _model_cls(g):
    class ExactGPModel(g.models.ExactGP):
        def __init__(self, train_inputs, train_targets, likelihood):
            super().__init__(train_inputs, train_targets, likelihood)
            self.mean_module = g.means.ConstantMean(constant_prior=g.priors.SmoothedBoxPrior(-1, 1))
            self.rbf_covar_module = g.kernels.RBFKernel(lengthscale_prior=g.priors.SmoothedBoxPrior(math.exp(-3), math.exp(3), sigma=0.1))
            self.covar_module = g.kernels.ScaleKernel(self.rbf_covar_module)

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    return ExactGPModel


def test_white_noise_regression_with_cg(dev):
    """test_white_noise_regression.py::test_posterior_latent_gp_and_likelihood_with_optimization_with_cg."""
    import gpytorch_amd as g
    from gpytorch_amd.likelihoods import GPInputWarning

    torch.manual_seed(1)
    train_x, test_x, train_y, test_y = _sine_data(dev)
    likelihood = g.likelihoods.FixedNoiseGaussianLikelihood(torch.ones(11, device=dev) * 0.001)
    gp_model = _model_cls(g)(train_x, train_y, likelihood).to(dev)
    mll = g.ExactMarginalLogLikelihood(likelihood, gp_model)
    gp_model.rbf_covar_module.initialize(lengthscale=math.exp(1))
    gp_model.mean_module.initialize(constant=0)
    gp_model.train()
    likelihood.train()
    optimizer = torch.optim.Adam(gp_model.parameters(), lr=0.1)
    with warnings.catch_warnings(), g.settings.debug(False), g.settings.max_cholesky_size(0):
        warnings.simplefilter("ignore", GPInputWarning)
        for _ in range(75):
            optimizer.zero_grad()
            output = gp_model(train_x)
            loss = -mll(output, train_y)
            loss.backward()
            optimizer.step()
        for param in gp_model.parameters():
            assert param.grad is not None
            assert param.grad.norm().item() > 0
        optimizer.step()
        gp_model.eval()
        likelihood.eval()
        test_function_predictions = likelihood(gp_model(test_x))
        mean_abs_error = torch.mean(torch.abs(test_y - test_function_predictions.mean))
    assert mean_abs_error.item() < 0.05, mean_abs_error.item()


def test_fast_pred_var_variance_within_5_percent(dev):
    """test_simple_gp_regression.py::test_posterior_latent_gp_and_likelihood_fast_pred_var, with CG / Lanczos forced."""
    import gpytorch_amd as g

    torch.manual_seed(1)
    train_x, test_x, train_y, test_y = _sine_data(dev)
    with g.settings.fast_pred_var(), g.settings.debug(False), g.settings.max_cholesky_size(0):
        likelihood = g.likelihoods.GaussianLikelihood(noise_prior=g.priors.SmoothedBoxPrior(math.exp(-3), math.exp(3), sigma=0.1))
        gp_model = _model_cls(g)(train_x, train_y, likelihood).to(dev)
        mll = g.ExactMarginalLogLikelihood(likelihood, gp_model)
        gp_model.rbf_covar_module.initialize(lengthscale=math.exp(1))
        gp_model.mean_module.initialize(constant=0)
        likelihood.initialize(noise=math.exp(1))
        gp_model.train()
        likelihood.train()
        optimizer = torch.optim.Adam(gp_model.parameters(), lr=0.1)
        for _ in range(50):
            optimizer.zero_grad()
            loss = -mll(gp_model(train_x), train_y)
            loss.backward()
            optimizer.step()
        for param in gp_model.parameters():
            assert param.grad is not None
            assert param.grad.norm().item() > 0
        gp_model.eval()
        likelihood.eval()
        _ = likelihood(gp_model(train_x))                       # set the caches
        likelihood.noise_covar.raw_noise.data.fill_(3)          # now a huge noise: variance ~ noise
        preds = likelihood(gp_model(train_x))
        noise = likelihood.noise_covar.noise
        var_diff = (preds.variance - noise).abs()
    assert float(torch.max(var_diff / noise)) < 0.05


def test_cg_solve_vs_dense_inverse_reference_size(dev):
    """test_lazy_evaluated_kernel_tensor.py::_test_inv_matmul at the reference's own size and tolerances: ``RBFKernel()``
    on ``randn(5, 6)`` (one member of its batch of two), ``kernel(x, x).solve(rhs)`` with ``max_cholesky_size(0)`` and
    ``cg_tolerance(1e-4)`` vs the dense inverse: rtol 0.02 / atol 1e-5, hyper-parameter gradients rtol 1e-3, rhs gradients
    rtol 0.03 / atol 1e-5, and linear_cg must have been called."""
    import gpytorch_amd as g
    from gpytorch_amd import linear_cg as LCG
    from oracle import kernels as OK

    torch.manual_seed(0)
    x = torch.randn(5, 6)
    rhs0 = torch.randn(5, 3)
    grad = torch.randn(5, 3)
    kern = g.kernels.RBFKernel().to(dev)
    rhs = rhs0.to(dev).requires_grad_(True)
    LCG.LAST_INFO = None
    with g.settings.max_cholesky_size(0), g.settings.cg_tolerance(1e-4):
        res = kern(x.to(dev), x.to(dev)).solve(rhs)
    assert LCG.LAST_INFO is not None and LCG.LAST_INFO.iterations > 0
    ls64 = kern.lengthscale.detach().double().cpu().requires_grad_(True)
    rhs64 = rhs0.double().requires_grad_(True)
    actual = torch.linalg.solve(OK.rbf(x.double(), x.double(), ls64, x1_eq_x2=True), rhs64)
    assert torch.allclose(res.detach().double().cpu(), actual.detach(), rtol=0.02, atol=1e-5)
    res.backward(gradient=grad.to(dev))
    actual.backward(gradient=grad.double())
    want = ls64.grad * torch.sigmoid(kern.raw_lengthscale.detach().double().cpu())
    assert torch.allclose(kern.raw_lengthscale.grad.double().cpu(), want, rtol=1e-3)
    assert torch.allclose(rhs.grad.double().cpu(), rhs64.grad, rtol=0.03, atol=1e-5)


@pytest.mark.parametrize("ard", [False, True])
def test_cg_solve_vs_dense_inverse_with_gradients(ard, dev):
    """The same check scaled up (n = 900, d = 5, ARD): solve through linear_cg at ``cg_tolerance(1e-4)`` against dense
    float64 autograd through the oracle's kernel.  The reference's atol of 1e-5 presumes CG converges exactly (n = 5); at
    n = 900 the solve error is bounded by the requested relative residual, so values are compared at rtol 0.02 with
    atol = 2e-4 max|x| (rhs gradients likewise); hyper-parameter gradients rtol 2e-3."""
    import gpytorch_amd as g
    from gpytorch_amd import linear_cg as LCG
    from oracle import kernels as OK

    n, d = 900, 5
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(n, d, generator=gen)
    rhs0 = torch.randn(n, 4, generator=gen)
    grad = torch.randn(n, 4, generator=gen)
    kern = g.kernels.RBFKernel(ard_num_dims=d if ard else None).to(dev)
    ls0 = (1.5 + 0.5 * torch.rand(1, d, generator=gen)) if ard else torch.tensor([[1.7]])
    kern.lengthscale = ls0.to(dev)
    xd = x.to(dev)
    rhs = rhs0.to(dev).requires_grad_(True)
    LCG.LAST_INFO = None
    with g.settings.max_cholesky_size(0), g.settings.cg_tolerance(1e-4), g.settings.max_preconditioner_size(0):
        op = kern(xd, xd).add_jitter(1.0)
        res = op.solve(rhs)
    assert LCG.LAST_INFO is not None and LCG.LAST_INFO.iterations > 0          # "linear_cg was called"
    # dense float64 ground truth with autograd
    ls64 = ls0.double().requires_grad_(True)
    rhs64 = rhs0.double().requires_grad_(True)
    K = OK.rbf(x.double(), x.double(), ls64, x1_eq_x2=True) + torch.eye(n, dtype=torch.float64)
    actual = torch.linalg.solve(K, rhs64)
    assert torch.allclose(res.detach().double().cpu(), actual.detach(), rtol=0.02, atol=2e-4 * float(actual.abs().max()))
    res.backward(gradient=grad.to(dev))
    actual.backward(gradient=grad.double())
    # d/d raw_lengthscale = d/d lengthscale * sigmoid(raw)
    chain = torch.sigmoid(kern.raw_lengthscale.detach().double().cpu())
    got = kern.raw_lengthscale.grad.double().cpu()
    want = ls64.grad * chain
    assert torch.allclose(got, want, rtol=2e-3, atol=1e-6 * float(want.abs().max())), (got, want)
    assert torch.allclose(rhs.grad.double().cpu(), rhs64.grad, rtol=0.03, atol=2e-4 * float(rhs64.grad.abs().max()))


def _simple_cases():
    from tests import simple_gp_cases as C

    return C.CASES + C.MULTITASK_CASES


@pytest.mark.parametrize("case", _simple_cases(), ids=[c.__name__ for c in _simple_cases()])
@pytest.mark.filterwarnings("ignore")
def test_simple_gp_regression_cases_of_the_reference(case, dev):
    """test/examples/test_simple_gp_regression.py:47-330, case by case (tests/simple_gp_cases.py), on the device."""
    import gpytorch_amd as g

    torch.manual_seed(1)
    case(g, dev)


@pytest.mark.parametrize("name", ["case_posterior_with_optimization", "case_skip_variances", "case_train_on_single_set_test_on_batch",
                                  "case_train_on_batch_shared_hypers_over_batch", "case_missing_data_single", "case_fantasy_updates"])
@pytest.mark.filterwarnings("ignore")
def test_simple_gp_regression_cases_through_the_bbmm_branch(name, dev):
    """The same reference cases with the dense branch switched off (``max_cholesky_size(0)``): every MLL, gradient, mean cache and LOVE cache of these
    11-to-41-point problems comes from mBCG / SLQ / Lanczos on the fused kernels (the reference does the same in its CG-forcing example tests,
    e.g. test/examples/test_white_noise_regression.py:56-102)."""
    import gpytorch_amd as g
    from tests import simple_gp_cases as C

    torch.manual_seed(1)
    with g.settings.max_cholesky_size(0), g.settings.num_trace_samples(100), g.settings.cg_tolerance(0.01), g.settings.max_preconditioner_size(0):
        getattr(C, name)(g, dev)
