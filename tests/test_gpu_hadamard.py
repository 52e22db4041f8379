"""GPU parity: Hadamard multitask GP (data kernel o IndexKernel looked up at per-point task indices) on the fused path.
Reference: gpytorch/kernels/index_kernel.py:101-112 and test/examples/test_hadamard_multitask_gp_regression.py:33-100
(MAE < 0.1 per task after training).  Ground truth: dense float64 K o B + torch autograd."""
import math

import pytest
import torch

from oracle import exact_gp as OG
from oracle import kernels as OK
from tests.util import rel_err

pytestmark = pytest.mark.gpu


def _model(g, x, i, y, dev):
    class HadamardMultitaskGPModel(g.models.ExactGP):
        def __init__(self, train_x, train_y, likelihood):
            super().__init__(train_x, train_y, likelihood)
            self.mean_module = g.means.ConstantMean()
            self.covar_module = g.kernels.RBFKernel()
            self.task_covar_module = g.kernels.IndexKernel(num_tasks=2, rank=1)

        def forward(self, x, i):
            covar_xi = self.covar_module(x).mul(self.task_covar_module(i))
            return g.distributions.MultivariateNormal(self.mean_module(x), covar_xi)

    lik = g.likelihoods.GaussianLikelihood().to(dev)
    return HadamardMultitaskGPModel((x.to(dev), i.to(dev)), y.to(dev), lik).to(dev), lik


@pytest.mark.parametrize("branch", ["cholesky", "bbmm"])
def test_hadamard_operator_mll_and_grads(branch, dev):
    import gpytorch_amd as g

    n = 400 if branch == "cholesky" else 1600
    gen = torch.Generator().manual_seed(0)
    x = torch.rand(n, 2, generator=gen)
    i = torch.randint(0, 2, (n,), generator=gen)
    y = torch.where(i == 0, torch.sin(4 * x[:, 0]), torch.cos(3 * x[:, 1])) + 0.1 * torch.randn(n, generator=gen)
    m, lik = _model(g, x, i, y, dev)
    Bf = torch.tensor([[0.9], [-0.4]])
    v = torch.tensor([0.3, 0.5])
    m.covar_module.lengthscale = 0.35
    with torch.no_grad():
        m.task_covar_module.covar_factor.copy_(Bf)
    m.task_covar_module.var = v
    lik.noise = 0.1
    op = lik(m.train()(*m.train_inputs)).lazy_covariance_matrix
    ktt = (Bf @ Bf.t() + torch.diag(v)).double()
    Kx = OK.rbf(x.double(), x.double(), 0.35, x1_eq_x2=True, direct=True)
    Kh = Kx * ktt[i][:, i] + 0.1 * torch.eye(n, dtype=torch.float64)
    V = torch.randn(n, 5, generator=gen)
    with torch.no_grad():
        assert rel_err(op @ V.to(dev), Kh @ V.double()) < 2e-5
        assert rel_err(op.diagonal(), Kh.diagonal()) < 1e-5
    mll = g.ExactMarginalLogLikelihood(lik, m)
    S = g.settings
    with S.max_cholesky_size(10_000 if branch == "cholesky" else 0), S.cg_tolerance(1e-5), S.num_trace_samples(400), S.max_preconditioner_size(0):
        torch.manual_seed(0)
        val = mll(m(*m.train_inputs), m.train_targets)
        val.backward()
    p_ls = torch.tensor(0.35, dtype=torch.float64, requires_grad=True)
    p_B = Bf.double().clone().requires_grad_(True)
    p_v = v.double().clone().requires_grad_(True)
    p_nz = torch.tensor(0.1, dtype=torch.float64, requires_grad=True)
    Kref = OK.rbf(x.double(), x.double(), p_ls, x1_eq_x2=True, direct=True) * (p_B @ p_B.t() + torch.diag(p_v))[i][:, i] + p_nz * torch.eye(n, dtype=torch.float64)
    ref = OG.dense_log_prob(Kref, y.double()) / n
    g_ls, g_B, g_v, g_nz = torch.autograd.grad(ref, [p_ls, p_B, p_v, p_nz])
    tol_v, tol_g = (2e-4, 3e-3) if branch == "cholesky" else (5e-3, 0.12)
    assert abs(float(val) - float(ref)) < tol_v * max(1.0, abs(float(ref)))
    sp = lambda t_: 1.0 - torch.exp(-t_)  # noqa: E731
    got_ls = float(m.covar_module.raw_lengthscale.grad.sum())
    assert abs(got_ls - float(g_ls) * (1 - math.exp(-0.35))) < tol_g * abs(float(g_ls) * (1 - math.exp(-0.35))) + 1e-5
    assert rel_err(m.task_covar_module.covar_factor.grad, g_B) < tol_g
    assert rel_err(m.task_covar_module.raw_var.grad, g_v * sp(v.double())) < tol_g


def test_hadamard_multitask_gp_mean_abs_error(dev):
    """test_hadamard_multitask_gp_regression.py::test_multitask_gp_mean_abs_error with CG forced: 100 Adam steps, MAE < 0.1 per task."""
    import gpytorch_amd as g

    torch.manual_seed(0)
    train_x = torch.linspace(0, 1, 100)
    i1, i2 = torch.zeros(100, dtype=torch.long), torch.ones(100, dtype=torch.long)
    y1 = torch.sin(train_x * (2 * math.pi)) + torch.randn(100) * 1e-2
    y2 = torch.cos(train_x * (2 * math.pi)) + torch.randn(100) * 1e-2
    test_x = torch.linspace(0, 1, 51)
    m, lik = _model(g, torch.cat([train_x, train_x]), torch.cat([i1, i2]), torch.cat([y1, y2]), dev)
    mll = g.ExactMarginalLogLikelihood(lik, m)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=0.05)
    with g.settings.max_cholesky_size(0), g.settings.debug(False), g.settings.num_trace_samples(20):
        for _ in range(100):
            opt.zero_grad()
            loss = -mll(m(*m.train_inputs), m.train_targets)
            loss.backward()
            opt.step()
        for p in m.parameters():
            assert p.grad is not None and p.grad.norm().item() > 0
        m.eval()
        lik.eval()
        with torch.no_grad():
            p1 = lik(m(test_x.to(dev), torch.zeros(51, dtype=torch.long, device=dev))).mean
            p2 = lik(m(test_x.to(dev), torch.ones(51, dtype=torch.long, device=dev))).mean
    assert float((torch.sin(test_x * 2 * math.pi).to(dev) - p1).abs().mean()) < 0.1
    assert float((torch.cos(test_x * 2 * math.pi).to(dev) - p2).abs().mean()) < 0.1


def test_rhs_refinement_on_the_hadamard_operator(dev):
    """Round 6: ``settings.rhs_refinement`` on the Hadamard (data kernel o task kernel of the observed tasks) operator: float64 residual through
    ``hadamard_matvec`` on the prepared points widened to float64, one more float32 solve of it.  Ill-conditioned (noise 1e-3): the float32 mBCG solve
    stalls near 1e-3 of the dense float64 solution; refined, it is an order of magnitude closer."""
    import gpytorch_amd as g

    n = 3000
    gen = torch.Generator().manual_seed(1)
    x = torch.rand(n, 2, generator=gen)
    i = torch.randint(0, 2, (n,), generator=gen)
    y = torch.where(i == 0, torch.sin(4 * x[:, 0]), torch.cos(3 * x[:, 1])) + 0.05 * torch.randn(n, generator=gen)
    m, lik = _model(g, x, i, y, dev)
    Bf = torch.tensor([[0.9], [-0.4]])
    v = torch.tensor([0.3, 0.5])
    m.covar_module.lengthscale = 0.35
    with torch.no_grad():
        m.task_covar_module.covar_factor.copy_(Bf)
    m.task_covar_module.var = v
    lik.noise = 1e-3
    op = lik(m.train()(*m.train_inputs)).lazy_covariance_matrix
    ktt = (Bf @ Bf.t() + torch.diag(v)).double()
    Kh = OK.rbf(x.double(), x.double(), 0.35, x1_eq_x2=True, direct=True) * ktt[i][:, i] + 1e-3 * torch.eye(n, dtype=torch.float64)
    assert op.float64_product_available()
    V = torch.randn(n, 3, generator=gen, dtype=torch.float64)
    assert rel_err(op.matmul_float64(V.to(dev)), Kh @ V) < 1e-5
    sol_ref = torch.linalg.solve(Kh, y.double().unsqueeze(-1))
    S = g.settings
    errs = {}
    for refine in (False, True):
        with torch.no_grad(), S.max_cholesky_size(0), S.cg_tolerance(1e-4), S.max_cg_iterations(4000), S.rhs_refinement(refine):
            errs[refine] = rel_err(op.solve(y.to(dev).unsqueeze(-1)), sol_ref)
    assert errs[True] < 0.2 * errs[False] and errs[True] < 2e-4, errs
