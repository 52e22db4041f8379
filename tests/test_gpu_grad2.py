"""GPU parity: the Gram-form fused bilinear derivative (csrc/kv_grad2.hpp) -- hyper-parameter sums AND input gradients --
against float64 autograd through the oracle's kernels, and the gradient checks of the reference's fused-kernel harness:

  * kernel-level: sum_c L_c^T K(x1, x2) R_c differentiated w.r.t. lengthscale(s), outputscale, x1, x2
    (what LinearOperator._bilinear_derivative + RBFCovariance.backward / MaternCovariance.backward compute,
    gpytorch/functions/rbf_covariance.py:26-29, matern_covariance.py:53-56; x-gradients as in
    test/lazy/test_lazy_evaluated_kernel_tensor.py:44-48: ``x1.grad + x2.grad``)
  * harness: ``kern(x, x).sum()`` hyper-parameter gradients, rtol 1e-3 / atol 1e-3
    (gpytorch/test/base_keops_test_case.py:105-132), incl. ARD
  * model-level: d MLL / d x through the BBMM branch and the Cholesky branch (deep-kernel style input gradients).
"""
import math

import pytest
import torch

from oracle import kernels as OK
from tests.util import make_data, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["rbf", "matern52", "matern32"])
@pytest.mark.parametrize("n,m,d,t,ard", [(333, 517, 5, 37, True), (700, 450, 3, 70, False), (130, 1000, 10, 11, True), (260, 129, 1, 3, False),
                                         (515, 515, 16, 66, True),
                                         # round 4: the per-dimension mode of the split W contraction beyond 6 dimensions (one wave per SIMD, no spills)
                                         (400, 600, 10, 40, True), (300, 500, 8, 65, True), (260, 300, 12, 30, True)])
def test_bilinear_derivative_and_input_gradients(kind, n, m, d, t, ard, dev):
    from gpytorch_amd import backend as B
    from gpytorch_amd.functions import hyper_grads

    if kind != "rbf" and d in (1, 16, 12):
        pytest.skip("dimension sweep is exhaustive for rbf")
    g0 = torch.Generator().manual_seed(n + m + t)
    X1 = torch.rand(n, d, generator=g0, dtype=torch.float64).requires_grad_(True)
    X2 = torch.rand(m, d, generator=g0, dtype=torch.float64).requires_grad_(True)
    Lm = torch.randn(n, t, generator=g0, dtype=torch.float64)
    Rm = torch.randn(m, t, generator=g0, dtype=torch.float64)
    base = 0.3 + 0.1 * d
    ls = (base + 0.2 * torch.rand(1, d if ard else 1, generator=g0, dtype=torch.float64)).requires_grad_(True)
    os_ = torch.tensor(1.7, dtype=torch.float64, requires_grad=True)
    K = OK.kernel_matrix(kind, X1, X2, ls, os_, x1_eq_x2=False, direct=True)
    val = (Lm * (K @ Rm)).sum()
    gl, go, gx1, gx2 = torch.autograd.grad(val, [ls, os_, X1, X2])
    shift = X1.detach().mean(0).float().to(dev)
    lsd = ls.detach().float().to(dev)
    p1 = B.prep_points(kind, X1.detach().float().to(dev), lsd, shift)
    p2 = B.prep_points(kind, X2.detach().float().to(dev), lsd, shift)
    assert B.grad_gram_ok(p1, p2)
    lt, rt = B.to_probe_major(Lm.to(dev)), B.to_probe_major(Rm.to(dev))
    osd = os_.detach().float().reshape(1).to(dev)
    d_ls, d_os, d_x1, d_x2 = hyper_grads(p1, p2, lsd, osd, lt, rt, want_x1=True, want_x2=True)
    assert rel_err(d_ls, gl) < 1e-3, (kind, d_ls, gl)
    assert abs(float(d_os) - float(go)) < 1e-3 * abs(float(go))
    assert rel_err(d_x1, gx1) < 1e-3
    assert rel_err(d_x2, gx2) < 1e-3
    # hyper-parameters only (single lengthscale -> the VALU-only mode) and the direct-difference fallback agree
    d_ls2, d_os2 = hyper_grads(p1, p2, lsd, osd, lt, rt)
    assert rel_err(d_ls2, gl) < 1e-3 and abs(float(d_os2) - float(go)) < 1e-3 * abs(float(go))
    try:
        B.FORCE_GRAD_DIRECT = True
        d_ls3, d_os3 = hyper_grads(p1, p2, lsd, osd, lt, rt)
    finally:
        B.FORCE_GRAD_DIRECT = False
    assert rel_err(d_ls3, gl) < 1e-3 and abs(float(d_os3) - float(go)) < 1e-3 * abs(float(go))


@pytest.mark.parametrize("ard", [False, True])
@pytest.mark.parametrize("kind", ["rbf", "matern52"])
def test_keops_harness_gradient(kind, ard, dev):
    """base_keops_test_case.py::test_gradient: d kern(x, x).sum() / d hyper-parameters vs the dense kernel, rtol = atol = 1e-3
    (one member of the reference's batch of four)."""
    import gpytorch_amd as g

    torch.manual_seed(0)
    x = torch.randn(100, 3)
    kw = dict(ard_num_dims=3) if ard else {}
    kern = (g.kernels.RBFKernel(**kw) if kind == "rbf" else g.kernels.MaternKernel(nu=2.5, **kw)).to(dev)
    s1 = kern(x.to(dev), x.to(dev)).sum()
    (grad1,) = torch.autograd.grad(s1, [kern.raw_lengthscale])
    ls64 = kern.lengthscale.detach().double().cpu().requires_grad_(True)
    s2 = OK.kernel_matrix(kind, x.double(), x.double(), ls64, 1.0, x1_eq_x2=True).sum()
    (grad2,) = torch.autograd.grad(s2, [ls64])
    grad2 = grad2 * torch.sigmoid(kern.raw_lengthscale.detach().double().cpu())
    assert abs(float(s1) - float(s2)) < 1e-4 * abs(float(s2))
    assert torch.allclose(grad1.double().cpu(), grad2, rtol=1e-3, atol=1e-3)


def test_kernel_matmul_input_gradients(dev):
    """test_lazy_evaluated_kernel_tensor.py::_test_matmul: (K @ rhs).backward(grad) gives kernel-parameter gradients (rtol
    1e-3) and ``x1.grad + x2.grad`` (rtol 1e-3) equal to the dense evaluation."""
    import gpytorch_amd as g

    torch.manual_seed(0)
    x = torch.randn(60, 6)
    rhs0 = torch.randn(60, 4)
    grad = torch.randn(60, 4)
    kern = g.kernels.RBFKernel().to(dev)
    x1 = x.to(dev).requires_grad_(True)
    x2 = x.to(dev).clone().requires_grad_(True)
    res = kern(x1, x2).matmul(rhs0.to(dev))
    res.backward(gradient=grad.to(dev))
    xa = x.double().requires_grad_(True)
    xb = x.double().clone().requires_grad_(True)
    ls64 = kern.lengthscale.detach().double().cpu().requires_grad_(True)
    actual = OK.rbf(xa, xb, ls64, x1_eq_x2=False, direct=True) @ rhs0.double()
    assert rel_err(res, actual) < 2e-5
    actual.backward(gradient=grad.double())
    want_ls = ls64.grad * torch.sigmoid(kern.raw_lengthscale.detach().double().cpu())
    assert torch.allclose(kern.raw_lengthscale.grad.double().cpu(), want_ls, rtol=1e-3)
    got = (x1.grad + x2.grad).double().cpu()
    want = xa.grad + xb.grad
    assert torch.allclose(got, want, rtol=1e-3, atol=1e-3 * float(want.abs().max()))


@pytest.mark.parametrize("branch", ["cholesky", "bbmm"])
def test_mll_gradient_wrt_train_inputs(branch, dev):
    """d MLL / d x (what deep kernel learning back-propagates into the feature extractor) vs dense float64 autograd: the full
    MLL on the Cholesky branch, the inverse quadratic form on the BBMM branch."""
    import gpytorch_amd as g
    from oracle import exact_gp as OG

    n, d, ls = (500, 3, 0.3) if branch == "cholesky" else (1500, 3, 0.3)
    X, y = make_data(n, d)

    class M(g.models.ExactGP):
        def __init__(self, x, yy, lik):
            super().__init__(x, yy, lik)
            self.mean_module = g.means.ConstantMean()
            self.covar_module = g.kernels.ScaleKernel(g.kernels.RBFKernel())

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    lik = g.likelihoods.GaussianLikelihood().to(dev)
    xd = X.float().to(dev).requires_grad_(True)
    m = M(xd, y.float().to(dev), lik).to(dev)
    m.covar_module.base_kernel.lengthscale = ls
    m.covar_module.outputscale = 1.2
    lik.noise = 0.1
    mll = g.ExactMarginalLogLikelihood(lik, m)
    m.train()
    lik.train()
    S = g.settings
    X64 = X.clone().requires_grad_(True)
    Kh = OK.kernel_matrix("rbf", X64, X64, ls, 1.2, x1_eq_x2=True, direct=True) + 0.1 * torch.eye(n, dtype=torch.float64)
    with S.max_cholesky_size(10_000 if branch == "cholesky" else 0), S.cg_tolerance(1e-5), S.max_preconditioner_size(0), S.debug(False):
        if branch == "cholesky":
            val = mll(m(xd), m.train_targets)
            ref = OG.dense_log_prob(Kh, y) / n
        else:
            # the inverse quadratic form alone is deterministic on the BBMM branch (the log-det gradient is a stochastic trace
            # estimate): y^T K_hat^-1 y through mBCG, its x-gradient through two fused derivative passes
            val, _ = lik(m(xd)).lazy_covariance_matrix.inv_quad_logdet(m.train_targets.unsqueeze(-1), logdet=False)
            ref = (y * torch.linalg.solve(Kh, y)).sum()
        (gx,) = torch.autograd.grad(val, [xd])
    (gref,) = torch.autograd.grad(ref, [X64])
    assert abs(float(val) - float(ref)) < 1e-3 * abs(float(ref))
    assert float((gx.double().cpu() - gref).norm() / gref.norm()) < 2e-3
    _ = math
