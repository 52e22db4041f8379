"""GPU parity: multitask exact GP on the fused Kronecker path (BASELINE config 5 in miniature) against dense
float64 Cholesky (oracle/multitask.py; the reference's own test asserts MAE < 0.05 per task,
test/examples/test_kronecker_multitask_gp_regression.py:89-90)."""
import math

import pytest
import torch

from oracle import multitask as OM
from tests.util import rel_err

pytestmark = pytest.mark.gpu
T = 3


def _data(n, d, seed=0):
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(n, d, generator=g, dtype=torch.float64)
    Y = torch.stack([torch.sin(2 * math.pi * X[:, 0]) + 0.3 * X[:, 1], torch.cos(2 * math.pi * X[:, 0]), torch.sin(math.pi * X.sum(-1))], -1)
    Y = Y + 0.1 * torch.randn(n, T, generator=g, dtype=torch.float64)
    return X, Y


def _model(X, Y, dev, ls=0.35, Bf=None, v=None, tn=None):
    import gpytorch_amd as g

    class MT(g.models.ExactGP):
        def __init__(self, x, y, lik):
            super().__init__(x, y, lik)
            self.mean_module = g.means.MultitaskMean(g.means.ZeroMean(), num_tasks=T)
            self.covar_module = g.kernels.MultitaskKernel(g.kernels.RBFKernel(), num_tasks=T, rank=1)

        def forward(self, x):
            return g.distributions.MultitaskMultivariateNormal(self.mean_module(x), self.covar_module(x))

    lik = g.likelihoods.MultitaskGaussianLikelihood(num_tasks=T, has_global_noise=False).to(dev)
    m = MT(X.float().to(dev), Y.float().to(dev), lik).to(dev)
    m.covar_module.data_covar_module.lengthscale = ls
    with torch.no_grad():
        m.covar_module.task_covar_module.covar_factor.copy_(Bf.float())
    m.covar_module.task_covar_module.var = v.float()
    lik.task_noises = tn.float()
    return g, m, lik


PARAMS = dict(Bf=torch.tensor([[0.9], [-0.5], [0.7]], dtype=torch.float64), v=torch.tensor([0.4, 0.6, 0.3], dtype=torch.float64),
              tn=torch.tensor([0.05, 0.1, 0.08], dtype=torch.float64))


def test_kronecker_matmul_and_diag(dev):
    import gpytorch_amd as g

    X, Y = _data(400, 2)
    g_, m, lik = _model(X, Y, dev, **PARAMS)
    op = lik(m.train()(m.train_inputs[0])).lazy_covariance_matrix
    Kh = OM.khat("rbf", X, 0.35, 1.0, PARAMS["Bf"], PARAMS["v"], PARAMS["tn"])
    V = torch.randn(400 * T, 5, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    with torch.no_grad():
        out = op @ V.float().to(dev)
    assert rel_err(out, Kh @ V) < 5e-5
    assert rel_err(op.diagonal(), Kh.diagonal()) < 1e-5
    assert rel_err(op.to_dense(), Kh) < 1e-5
    _ = g


@pytest.mark.parametrize("branch", ["cholesky", "bbmm"])
def test_multitask_mll_and_grads(branch, dev):
    n = 200 if branch == "cholesky" else 160
    X, Y = _data(n, 2)
    g, m, lik = _model(X, Y, dev, **PARAMS)
    mll = g.ExactMarginalLogLikelihood(lik, m)
    m.train(); lik.train()
    S = g.settings
    # bbmm branch: a COMPLETE probe basis sqrt(N) e_k makes the Hutchinson trace exact, so the A.6 backward on the
    # Kronecker operator can be compared with the dense autograd gradient deterministically
    Z = math.sqrt(n * T) * torch.eye(n * T, dtype=torch.float64)
    S.deterministic_probes.probe_vectors = Z
    try:
        with S.max_cholesky_size(10_000 if branch == "cholesky" else 0), S.cg_tolerance(1e-4), S.deterministic_probes(True):
            val = mll(m(m.train_inputs[0]), m.train_targets)
            val.backward()
    finally:
        S.deterministic_probes.probe_vectors = None
    ref, gref = OM.dense_mll_and_grads("rbf", X, Y, 0.35, 1.0, PARAMS["Bf"], PARAMS["v"], PARAMS["tn"])
    tol_v, tol_g = (2e-4, 2e-3) if branch == "cholesky" else (5e-3, 2e-2)  # bbmm: 20-step SLQ value, exact-trace gradient
    assert abs(float(val) - float(ref)) < tol_v * max(1.0, abs(float(ref)))
    sg = lambda a: 1.0 - torch.exp(-a)  # noqa: E731  d softplus / d raw at the given actual value
    got_ls = float(m.covar_module.data_covar_module.raw_lengthscale.grad.sum())
    exp_ls = float(gref[0]) * (1 - math.exp(-0.35))
    assert abs(got_ls - exp_ls) < tol_g * abs(exp_ls) + 1e-4, (got_ls, exp_ls)
    got_B = m.covar_module.task_covar_module.covar_factor.grad.double().cpu()
    assert (got_B - gref[1]).abs().max() < tol_g * gref[1].abs().max() + 1e-4
    got_v = m.covar_module.task_covar_module.raw_var.grad.double().cpu()
    assert (got_v - gref[2] * sg(PARAMS["v"])).abs().max() < tol_g * (gref[2] * sg(PARAMS["v"])).abs().max() + 1e-4
    got_n = lik.raw_task_noises.grad.double().cpu()
    exp_n = gref[3] * sg(PARAMS["tn"] - 1e-4)
    assert (got_n - exp_n).abs().max() < tol_g * exp_n.abs().max() + 1e-4


@pytest.mark.parametrize("fast", [False, True])
def test_multitask_posterior(fast, dev):
    n, ns = 700, 60
    X, Y = _data(n, 2)
    Xs, _ = _data(ns, 2, seed=5)
    g, m, lik = _model(X, Y, dev, **PARAMS)
    m.eval(); lik.eval()
    S = g.settings
    torch.manual_seed(0)
    with torch.no_grad(), S.max_cholesky_size(0), S.eval_cg_tolerance(1e-4), S.fast_pred_var(fast), S.max_root_decomposition_size(400):
        pred = lik(m(Xs.float().to(dev)))
        mu, var = pred.mean, pred.variance
    mu_ref, var_ref = OM.dense_posterior("rbf", X, Y, Xs, 0.35, 1.0, PARAMS["Bf"], PARAMS["v"], PARAMS["tn"])
    assert mu.shape == (ns, T) and var.shape == (ns, T)
    assert rel_err(mu, mu_ref) < 2e-3
    assert ((var.double().cpu() - var_ref).abs() / var_ref).max() < (0.05 if fast else 5e-3)
    # reference's own criterion: MAE < 0.05 per task against the noiseless truth is data dependent; here: vs dense posterior
    assert float((mu.double().cpu() - mu_ref).abs().mean()) < 0.05


def test_multitask_posterior_with_rhs_refinement(dev):
    """Round 6: ``settings.rhs_refinement`` on the STRUCTURED operator -- the mean-cache solve refined through the float64 Kronecker MVM
    (``KroneckerFusedAddedDiagLinearOperator._matvec64``) and the exact predictive covariance from ``bbmm.variational_inv_quad`` (one float64
    product, no second solve): the variance of F (without the task noise) against dense float64, which the unrefined float32 path misses."""
    n, ns = 900, 40
    X, Y = _data(n, 2)
    Xs, _ = _data(ns, 2, seed=5)
    g, m, lik = _model(X, Y, dev, **PARAMS)
    S = g.settings
    mu_ref, fvar_ref = OM.dense_posterior("rbf", X, Y, Xs, 0.35, 1.0, PARAMS["Bf"], PARAMS["v"], PARAMS["tn"], noise=False)
    out = {}
    for refine in (False, True):
        m.train(); m.eval(); lik.eval()
        with torch.no_grad(), S.max_cholesky_size(0), S.eval_cg_tolerance(1e-4), S.fast_pred_var(False), S.rhs_refinement(refine):
            pred = m(Xs.float().to(dev))
            out[refine] = (pred.mean.double().cpu(), pred.variance.double().cpu())
    err = {r: float(((out[r][1] - fvar_ref).abs() / (2e-3 * fvar_ref + 2e-6)).max()) for r in out}
    assert rel_err(out[True][0], mu_ref) < 2e-4 and rel_err(out[True][0], mu_ref) <= rel_err(out[False][0], mu_ref) + 1e-6
    assert err[True] < 1.0, err                       # the variance of f to rtol 2e-3 (+ the float32-input floor 2e-6)
    assert err[True] < err[False], err
