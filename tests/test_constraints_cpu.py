"""``test/constraints/test_constraints.py:28-240`` restated over ``gpytorch_amd.constraints`` (transform / inverse of every constraint class with float and
tensor bounds, un-enforced constraints with an initial value, the non-finite-bounds error, constraint lookup by dotted parameter name)."""
import math

import pytest
import torch
from torch import sigmoid
from torch.nn.functional import softplus

import gpytorch_amd as g

C = g.constraints
close = lambda a, b: torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)  # noqa: E731


def test_interval():
    c = C.Interval(1.0, 5.0)
    v = torch.tensor(-3.0)
    close(c.transform(v), (5.0 - 1.0) * sigmoid(v) + 1.0)
    close(c.inverse_transform(c.transform(v)), v)
    c = C.Interval(torch.tensor([1.0, 2.0]), torch.tensor([3.0, 4.0]))
    v = torch.tensor([-3.0, -2.0])
    close(c.transform(v), torch.stack([(3.0 - 1.0) * sigmoid(v[0]) + 1.0, (4.0 - 2.0) * sigmoid(v[1]) + 2.0]))
    close(c.inverse_transform(c.transform(v)), v)
    lo, hi = c                                    # (a constraint iterates over its two bounds)
    assert torch.equal(lo, c.lower_bound) and torch.equal(hi, c.upper_bound)


def test_initial_value_of_an_unenforced_constraint():
    c = C.Interval(1.0, 5.0, transform=None, initial_value=3.0)
    assert not c.enforced
    assert g.likelihoods.GaussianLikelihood(noise_constraint=c).noise.item() == 3.0
    # an enforced one: the raw parameter starts at the inverse transform of the initial value
    lk = g.likelihoods.GaussianLikelihood(noise_constraint=C.GreaterThan(1e-4, initial_value=0.25))
    assert abs(lk.noise.item() - 0.25) < 1e-6


def test_error_on_infinite_bounds():
    for bounds in ((0.0, math.inf), (-math.inf, 0.0)):
        with pytest.raises(ValueError, match="Cannot make an Interval directly with non-finite bounds"):
            C.Interval(*bounds)
    with pytest.raises(ValueError, match="empty intervals"):
        C.Interval(2.0, 1.0)


@pytest.mark.parametrize("kind", ["greater_than", "less_than", "positive"])
def test_one_sided_constraints(kind):
    make, expect = {
        "greater_than": (lambda b: C.GreaterThan(b), lambda v, b: softplus(v) + b),
        "less_than": (lambda b: C.LessThan(b), lambda v, b: -softplus(-v) + b),
        "positive": (lambda b: C.Positive(), lambda v, b: softplus(v)),
    }[kind]
    c, v = make(1.0), torch.tensor(-3.0)
    close(c.transform(v), expect(v, 1.0))
    close(c.inverse_transform(c.transform(v)), v)
    c, v = make(torch.tensor([1.0, 2.0])), torch.tensor([-3.0, -2.0])
    close(c.transform(v), expect(v, torch.tensor([1.0, 2.0])))
    close(c.inverse_transform(c.transform(v)), v)


def test_constraint_lookup_by_parameter_name():
    class ExactGPModel(g.models.ExactGP):
        def __init__(self, x, y, likelihood):
            super().__init__(x, y, likelihood)
            self.mean_module = g.means.ConstantMean()
            self.covar_module = g.kernels.ScaleKernel(g.kernels.RBFKernel())

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    model = ExactGPModel(None, None, g.likelihoods.GaussianLikelihood())
    assert isinstance(model.constraint_for_parameter_name("likelihood.noise_covar.raw_noise"), C.GreaterThan)
    assert isinstance(model.constraint_for_parameter_name("covar_module.base_kernel.raw_lengthscale"), C.Positive)
    seen = {name: constraint for name, _, constraint in model.named_parameters_and_constraints()}
    assert isinstance(seen["likelihood.noise_covar.raw_noise"], C.GreaterThan)
    assert isinstance(seen["covar_module.raw_outputscale"], C.Positive) and isinstance(seen["covar_module.base_kernel.raw_lengthscale"], C.Positive)
    mean_names = [n for n in seen if n.startswith("mean_module")]
    assert mean_names and all(seen[n] is None for n in mean_names)
