"""Dual-mode boundary (gpytorch_amd/dropin.py).

  * with the real ``gpytorch`` + ``linear_operator`` importable: the plugin kernel drops in under the reference's own
    ``ExactMarginalLogLikelihood`` / ``ExactGP`` (needs a GPU; skipped otherwise -- neither package can be installed in the build
    container);
  * always (CPU): the plugin classes are built against minimal stand-ins of the two packages and really subclass THEIR
    ``LinearOperator`` / ``Kernel``, forward every tensor to ``super().__init__`` (``representation()``), and take over ``+ noise``.
"""
import types

import pytest
import torch

from gpytorch_amd import dropin


def _stand_ins():
    class LinearOperator:
        def __init__(self, *args, **kwargs):
            self._args, self._kwargs = args, kwargs

        def representation(self):
            return tuple(a for a in self._args if torch.is_tensor(a))

        @property
        def shape(self):
            return self._size()

        @property
        def is_square(self):
            return self.shape[-1] == self.shape[-2]

        def add_diagonal(self, diag):
            raise NotImplementedError

        def __add__(self, other):
            raise NotImplementedError

    class ConstantDiagLinearOperator(LinearOperator):
        def __init__(self, diag_values, diag_shape):
            super().__init__(diag_values, diag_shape=diag_shape)
            self.diag_values, self.diag_shape = diag_values, diag_shape

    class RootLinearOperator(LinearOperator):
        def __init__(self, root):
            super().__init__(root)
            self.root = root

    class Kernel(torch.nn.Module):
        has_lengthscale = False

        def __init__(self, **kwargs):
            super().__init__()
            self.raw_lengthscale = torch.nn.Parameter(torch.zeros(1, 1))

        @property
        def lengthscale(self):
            return torch.nn.functional.softplus(self.raw_lengthscale)

    lo = types.SimpleNamespace(operators=types.SimpleNamespace(LinearOperator=LinearOperator, ConstantDiagLinearOperator=ConstantDiagLinearOperator,
                                                               RootLinearOperator=RootLinearOperator), to_linear_operator=lambda t: t)
    gp = types.SimpleNamespace(kernels=types.SimpleNamespace(Kernel=Kernel), mlls=types.SimpleNamespace(ExactMarginalLogLikelihood=object))
    return gp, lo


def test_plugin_classes_subclass_the_host_packages():
    gp, lo = _stand_ins()
    ns = dropin.build(gp, lo)
    assert issubclass(ns.FusedKernelOperator, lo.operators.LinearOperator)
    assert issubclass(ns.FusedAddedDiagOperator, lo.operators.LinearOperator)
    assert issubclass(ns.RBFKernel, gp.kernels.Kernel) and issubclass(ns.MaternKernel, gp.kernels.Kernel)
    assert ns.ExactMarginalLogLikelihood is gp.mlls.ExactMarginalLogLikelihood      # the reference's own class, untouched
    kern = ns.RBFKernel()
    x = torch.rand(7, 3)
    op = kern.forward(x, x)
    assert isinstance(op, ns.FusedKernelOperator) and op.shape == torch.Size([7, 7])
    rep = op.representation()
    assert any(r is x for r in rep) and any(r.shape == (1, 1) for r in rep)           # inputs + lengthscale reach representation()
    noisy = op + lo.operators.ConstantDiagLinearOperator(torch.tensor([0.1]), 7)
    assert isinstance(noisy, ns.FusedAddedDiagOperator)
    assert isinstance(op._transpose_nonbatch(), ns.FusedKernelOperator)
    with pytest.raises(RuntimeError, match="nu expected"):
        ns.MaternKernel(nu=1.0)


@pytest.mark.gpu
@pytest.mark.skipif(not dropin.AVAILABLE, reason="gpytorch / linear_operator are not importable in this environment")
def test_plugin_under_the_real_gpytorch(dev):  # pragma: no cover - runs only where the reference is installed
    import gpytorch

    from oracle import exact_gp as OG
    from tests.util import make_data

    n = 1200
    X, y = make_data(n, 3)

    class GPModel(gpytorch.models.ExactGP):
        def __init__(self, x, yy, lik):
            super().__init__(x, yy, lik)
            self.mean_module = gpytorch.means.ZeroMean()
            self.covar_module = gpytorch.kernels.ScaleKernel(dropin.RBFKernel())

        def forward(self, x):
            return gpytorch.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    lik = gpytorch.likelihoods.GaussianLikelihood().to(dev)
    m = GPModel(X.float().to(dev), y.float().to(dev), lik).to(dev)
    m.covar_module.base_kernel.lengthscale = 0.25
    m.covar_module.outputscale = 1.0
    lik.noise = 0.1
    mll = dropin.ExactMarginalLogLikelihood(lik, m)
    m.train()
    lik.train()
    with gpytorch.settings.max_cholesky_size(0), gpytorch.settings.num_trace_samples(200), gpytorch.settings.cg_tolerance(1e-4):
        val = mll(m(X.float().to(dev)), y.float().to(dev))
    ref, _ = OG.dense_mll_and_grads("rbf", X, y, 0.25, 1.0, 0.1)
    assert abs(float(val) - float(ref)) < 1e-2 * max(1.0, abs(float(ref)))
