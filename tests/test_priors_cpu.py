"""Priors: closed-form log-densities, the transform hook (priors/prior.py:26-34), and the
registry walk ``named_priors`` that ExactMarginalLogLikelihood sums over (mlls/exact_marginal_log_likelihood.py:41-52).
No GPU needed: the kernel / likelihood modules are only constructed, never evaluated."""
import math

import torch

import gpytorch_amd as g
from gpytorch_amd import priors as P


def test_log_densities_and_transform():
    x = torch.tensor(1.3)
    assert abs(float(P.GammaPrior(2.0, 0.5).log_prob(x)) - float(torch.distributions.Gamma(2.0, 0.5).log_prob(x))) < 1e-7
    assert abs(float(P.NormalPrior(0.2, 2.0).log_prob(x)) - (-0.5 * ((1.3 - 0.2) / 2.0) ** 2 - math.log(2.0) - 0.5 * math.log(2 * math.pi))) < 1e-6
    assert abs(float(P.LogNormalPrior(0.1, 0.7).log_prob(x)) - float(torch.distributions.LogNormal(0.1, 0.7).log_prob(x))) < 1e-7
    assert abs(float(P.UniformPrior(0.0, 2.0).log_prob(x)) + math.log(2.0)) < 1e-7
    assert abs(float(P.HalfCauchyPrior(1.5).log_prob(x)) - float(torch.distributions.HalfCauchy(1.5).log_prob(x))) < 1e-7
    assert abs(float(P.HalfNormalPrior(1.5).log_prob(x)) - float(torch.distributions.HalfNormal(1.5).log_prob(x))) < 1e-7
    # transform: the prior is placed on f(parameter)
    pt = P.NormalPrior(0.0, 1.0, transform=torch.log)
    assert abs(float(pt.log_prob(x)) - float(torch.distributions.Normal(0.0, 1.0).log_prob(x.log()))) < 1e-7
    # smoothed box: flat inside, Gaussian tails, normalised
    sb = P.SmoothedBoxPrior(0.1, 2.0, sigma=0.05)
    inside = -math.log(1.9 + math.sqrt(2 * math.pi) * 0.05)
    assert abs(float(sb.log_prob(torch.tensor(1.0))) - inside) < 1e-6
    assert abs(float(sb.log_prob(torch.tensor(2.1))) - (inside - 0.5 * (0.1 / 0.05) ** 2)) < 1e-5
    xs = torch.linspace(-1, 3, 200001)
    assert abs(float(torch.trapz(sb.log_prob(xs).exp(), xs)) - 1.0) < 1e-4


def test_priors_are_modules_and_registered():
    k = g.kernels.ScaleKernel(g.kernels.RBFKernel(lengthscale_prior=P.GammaPrior(3.0, 6.0)), outputscale_prior=P.GammaPrior(2.0, 0.15))
    lik = g.likelihoods.GaussianLikelihood(noise_prior=P.SmoothedBoxPrior(1e-3, 1.0, sigma=0.01))
    names = sorted(n for n, *_ in k.named_priors()) + sorted(n for n, *_ in lik.named_priors())
    assert names == ["base_kernel.lengthscale_prior", "outputscale_prior", "noise_covar.noise_prior"]
    assert "base_kernel.lengthscale_prior.concentration" in k.state_dict()
    k.base_kernel.lengthscale = 0.5
    total = sum(float(pr.log_prob(cl(mod)).sum()) for _, mod, pr, cl, _ in k.named_priors())
    want = float(torch.distributions.Gamma(3.0, 6.0).log_prob(torch.tensor(0.5))) + float(
        torch.distributions.Gamma(2.0, 0.15).log_prob(k.outputscale.detach()))
    assert abs(total - want) < 1e-5
