"""GPU: callers either side of the hot path (SURVEY.md section 8f / section 4 "examples" tests):
heteroskedastic fixed noise in the fused epilogue, an end-to-end hyper-parameter training loop with CG
forced (shape of test/examples/test_keops_gp_regression.py:43-77 and test_white_noise_regression.py:79-102),
and the probe-sharded multi-rank path on the REAL device code (two ranks sharing cuda:0 over gloo)."""
import math
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

from oracle import exact_gp as OG
from oracle import kernels as OK
from tests.util import make_data, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make_model(g, X, y, lik, dev, kind="rbf"):
    class GPModel(g.models.ExactGP):
        def __init__(self, x, yy, l):
            super().__init__(x, yy, l)
            self.mean_module = g.means.ConstantMean()
            base = g.kernels.RBFKernel() if kind == "rbf" else g.kernels.MaternKernel(nu=2.5)
            self.covar_module = g.kernels.ScaleKernel(base)

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    return GPModel(X.float().to(dev), y.float().to(dev), lik).to(dev)


def test_fixed_heteroskedastic_noise(dev):
    """FixedNoiseGaussianLikelihood with a per-point noise vector rides in the fused K_hat epilogue (dvec):
    solve / log-det / predictions against dense float64."""
    import gpytorch_amd as g

    n, ns, d, ls = 1400, 100, 3, 0.3
    X, y = make_data(n, d)
    Xs, _ = make_data(ns, d, seed=3)
    noise = 0.05 + 0.25 * torch.rand(n, generator=torch.Generator().manual_seed(9), dtype=torch.float64)
    lik = g.likelihoods.FixedNoiseGaussianLikelihood(noise.float().to(dev))
    m = _make_model(g, X, y, lik, dev)
    m.covar_module.base_kernel.lengthscale = ls
    m.covar_module.outputscale = 1.1
    Kh = OK.kernel_matrix("rbf", X, X, ls, 1.1, x1_eq_x2=True) + torch.diag(noise)
    S = g.settings
    # solve through the operator
    op = lik(m.train().__call__(m.train_inputs[0])).lazy_covariance_matrix
    assert type(op).__name__ == "FusedKernelAddedDiagLinearOperator" and op.noise_vec is not None
    with torch.no_grad(), S.max_cholesky_size(0), S.cg_tolerance(1e-4):
        sol = op.solve(y.float().to(dev).unsqueeze(-1))
    assert rel_err(sol, torch.linalg.solve(Kh, y.unsqueeze(-1))) < 1e-3
    assert rel_err(op.diagonal(), Kh.diagonal()) < 1e-5
    # MLL value with injected probes vs exact (SLQ accuracy) and Cholesky branch exactness
    mll = g.ExactMarginalLogLikelihood(lik, m)
    exact = OG.dense_log_prob(Kh, y - 0.0) / n
    with S.max_cholesky_size(10_000):
        v_chol = mll(m(m.train_inputs[0]), m.train_targets)
    assert abs(float(v_chol) - float(exact)) < 2e-4
    torch.manual_seed(0)
    with S.max_cholesky_size(0), S.cg_tolerance(1e-3), S.num_trace_samples(64):
        v_cg = mll(m(m.train_inputs[0]), m.train_targets)
    assert abs(float(v_cg) - float(exact)) < 0.02
    # posterior
    m.eval()
    with torch.no_grad(), S.max_cholesky_size(0), S.eval_cg_tolerance(1e-4):
        pred = m(Xs.float().to(dev))
    Ksx = OK.kernel_matrix("rbf", Xs, X, ls, 1.1, x1_eq_x2=False)
    mu_ref = Ksx @ torch.linalg.solve(Kh, y)
    var_ref = 1.1 - (Ksx * torch.linalg.solve(Kh, Ksx.t()).t()).sum(-1)
    assert rel_err(pred.mean, mu_ref) < 1e-3
    # latent (noise-free) variances are as small as 5e-3 here: the eval-CG tolerance (1e-4 relative to the prior
    # variance 1.1) bounds the ABSOLUTE error
    assert (pred.variance.double().cpu() - var_ref).abs().max() < 5e-4


def test_training_loop_with_cg_forced(dev):
    """25 Adam steps on the BBMM path (max_cholesky_size(0)): the loss decreases and the fitted model predicts
    the test function (reference: MAE < 0.15 on the KeOps example, test_keops_gp_regression.py:77)."""
    import gpytorch_amd as g

    torch.manual_seed(0)
    n = 900
    gen = torch.Generator().manual_seed(0)
    X = torch.rand(n, 2, generator=gen)
    y = torch.sin(4 * X[:, 0] * math.pi) * torch.cos(2 * X[:, 1]) + 0.1 * torch.randn(n, generator=gen)
    Xs = torch.rand(300, 2, generator=gen)
    ys = torch.sin(4 * Xs[:, 0] * math.pi) * torch.cos(2 * Xs[:, 1])
    lik = g.likelihoods.GaussianLikelihood().to(dev)
    m = _make_model(g, X, y, lik, dev)
    mll = g.ExactMarginalLogLikelihood(lik, m)
    opt = torch.optim.Adam(list(m.parameters()), lr=0.1)  # likelihood parameters are reachable through m.likelihood
    S = g.settings
    losses = []
    m.train()
    lik.train()
    with S.max_cholesky_size(0), S.num_trace_samples(16):
        for _ in range(25):
            opt.zero_grad()
            loss = -mll(m(m.train_inputs[0]), m.train_targets)
            loss.backward()
            opt.step()
            losses.append(float(loss))
    assert losses[-1] < losses[0] - 0.3, losses
    assert all(math.isfinite(v) for v in losses)
    m.eval()
    lik.eval()
    with torch.no_grad(), S.max_cholesky_size(0), S.fast_pred_var():
        pred = lik(m(Xs.to(dev)))
    mae = float((pred.mean.cpu() - ys).abs().mean())
    assert mae < 0.15, mae
    assert bool((pred.variance > 0).all())


def _rank_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from gpytorch_amd import backend as B
    from gpytorch_amd import distributed as D
    from gpytorch_amd.bbmm import inv_quad_logdet_forward

    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    n, t_total = 3000, 16
    X, y = make_data(n, 3)
    Z = torch.randn(n, t_total, generator=torch.Generator().manual_seed(1234), dtype=torch.float64)
    a, b = D.probe_shard(t_total, world, rank)
    xp = B.prep_points("rbf", X.float().to(dev), torch.tensor(0.25), X.mean(0).to(dev))
    sc, s2 = torch.tensor([1.0], device=dev), torch.tensor([0.1], device=dev)
    res = inv_quad_logdet_forward(xp, sc, s2, B.to_probe_major(y.unsqueeze(-1).to(dev)), precond=None, probes=Z[:, a:b],
                                  tolerance=0.5, group=dist.group.WORLD, t_total=t_total)
    q.put((rank, res.info.iterations, float(res.inv_quad.sum()), float(res.logdet)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_probe_sharding_on_device(dev):
    """Two processes share cuda:0 (gloo carries the 2-float / scalar all-reduces): the sharded MLL terms equal
    the single-process ones computed with all probes and the y column replicated `world` times."""
    from gpytorch_amd import backend as B
    from gpytorch_amd.bbmm import inv_quad_logdet_forward

    world, port = 2, 29500 + os.getpid() % 400
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=300) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n, t_total = 3000, 16
    X, y = make_data(n, 3)
    Z = torch.randn(n, t_total, generator=torch.Generator().manual_seed(1234), dtype=torch.float64)
    xp = B.prep_points("rbf", X.float().to(dev), torch.tensor(0.25), X.mean(0).to(dev))
    sc, s2 = torch.tensor([1.0], device=dev), torch.tensor([0.1], device=dev)
    rhs = B.to_probe_major(y.unsqueeze(-1).repeat(1, world).to(dev))
    ref = inv_quad_logdet_forward(xp, sc, s2, rhs, precond=None, probes=Z, tolerance=0.5)
    for rank, iters, iq, ld in results:
        assert iters == ref.info.iterations
        assert abs(iq - float(ref.inv_quad[0])) < 1e-5 * abs(float(ref.inv_quad[0]))
        assert abs(ld - float(ref.logdet)) < 1e-5 * abs(float(ref.logdet))
