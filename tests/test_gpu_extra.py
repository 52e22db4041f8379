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
from tests.util import free_port, make_data, rel_err

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
    with torch.no_grad(), S.max_cholesky_size(0), S.cg_tolerance(3e-5):   # (at 1e-4 the solve error sits AT the 1e-3 bound: 0.97e-3 .. 1.02e-3 depending on the summation order of the one-column kernel)
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


@pytest.mark.parametrize("branch", ["cholesky", "bbmm"])
def test_fixed_noise_learn_additional_noise_gets_a_gradient(branch, dev):
    """FixedNoiseGaussianLikelihood(learn_additional_noise=True) with a NON-constant fixed noise vector (gaussian_likelihood.py:337-352):
    the learned second noise stays a differentiable scalar beside the fixed vector (FixedPlusConstantDiagLinearOperator), so the MLL has a
    gradient w.r.t. second_noise_covar.raw_noise -- value and gradient against dense float64 autograd."""
    import gpytorch_amd as g

    n, d, ls = 900, 2, 0.3
    X, y = make_data(n, d)
    fixed = 0.05 + 0.2 * torch.rand(n, generator=torch.Generator().manual_seed(4), dtype=torch.float64)
    lik = g.likelihoods.FixedNoiseGaussianLikelihood(fixed.float().to(dev), learn_additional_noise=True).to(dev)
    lik.second_noise = 0.07
    m = _make_model(g, X, y, lik, dev)
    m.covar_module.base_kernel.lengthscale = ls
    m.covar_module.outputscale = 1.2
    m.mean_module.initialize(constant=0.0)
    m.train()
    lik.train()
    op = lik(m(m.train_inputs[0])).lazy_covariance_matrix
    assert op.noise_vec is not None and op.noise.requires_grad
    mll = g.ExactMarginalLogLikelihood(lik, m)
    S = g.settings
    torch.manual_seed(0)
    with S.max_cholesky_size(10_000 if branch == "cholesky" else 0), S.cg_tolerance(1e-4), S.num_trace_samples(256), S.max_preconditioner_size(0):
        val = mll(m(m.train_inputs[0]), m.train_targets)
        val.backward()
    s2 = torch.tensor(0.07, dtype=torch.float64, requires_grad=True)
    Kh = OK.kernel_matrix("rbf", X, X, ls, 1.2, x1_eq_x2=True) + torch.diag(fixed) + s2 * torch.eye(n, dtype=torch.float64)
    ref = OG.dense_log_prob(Kh, y) / n
    (gref,) = torch.autograd.grad(ref, s2)
    graw = lik.second_noise_covar.raw_noise.grad
    assert graw is not None and float(graw.abs().sum()) > 0
    chain = 1.0 - math.exp(-(0.07 - 1e-4))   # d softplus / d raw at noise = 0.07 (GreaterThan(1e-4) constraint)
    want = float(gref) * chain
    tol_v, tol_g = (2e-4, 3e-3) if branch == "cholesky" else (2e-2, 0.1)
    assert abs(float(val) - float(ref)) < tol_v * max(1.0, abs(float(ref)))
    assert abs(float(graw.sum()) - want) < tol_g * abs(want), (float(graw.sum()), want)


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
                                  tolerance=1e-3, group=dist.group.WORLD, t_total=t_total)
    q.put((rank, res.info.iterations, float(res.inv_quad.sum()), float(res.logdet)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_probe_sharding_on_device(dev):
    """Two processes share cuda:0 (gloo carries the 2-float / scalar all-reduces and the broadcast of the y solve): the
    sharded MLL terms equal the single-process ones computed with all probes + y (the y column lives on rank 0 only, so the
    global mean-residual stopping rule averages over exactly the same t_total + 1 columns)."""
    from gpytorch_amd import backend as B
    from gpytorch_amd.bbmm import inv_quad_logdet_forward

    world, port = 2, free_port()
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
    rhs = B.to_probe_major(y.unsqueeze(-1).to(dev))
    ref = inv_quad_logdet_forward(xp, sc, s2, rhs, precond=None, probes=Z, tolerance=1e-3)
    for rank, iters, iq, ld in results:
        # 8 probes (+ y on rank 0) per rank and the 17-column reference run on different kernels: different float32
        # summation orders move the stopping iteration by a few steps (see test_gpu_bbmm); the converged values agree
        assert abs(iters - ref.info.iterations) <= max(2, 0.03 * ref.info.iterations)
        assert abs(iq - float(ref.inv_quad[0])) < 1e-4 * abs(float(ref.inv_quad[0]))
        assert abs(ld - float(ref.logdet)) < 1e-4 * abs(float(ref.logdet))


def _collect(q, procs, world, budget=240.0):
    """Results of all ranks; fails within seconds when a rank dies instead of waiting out the queue timeout."""
    import queue
    import time

    results, t0 = [], time.time()
    while len(results) < world:
        try:
            results.append(q.get(timeout=2.0))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > budget:
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                raise AssertionError(f"rank process failed (exit codes {dead}) or timed out")
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(results, key=lambda r: r[0])


def _row_worker(rank, world, port, q, fixed_noise, precond=0, refine=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    import gpytorch_amd as g

    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    mu, var, info = _posterior(g, dev, fixed_noise, dist.group.WORLD, precond, refine)
    q.put((rank, mu.cpu().numpy(), var.cpu().numpy(), info))
    dist.barrier()
    dist.destroy_process_group()


def _posterior(g, dev, fixed_noise, row_group, precond=0, refine=False):
    n, ns, d = 1501, 257, 3  # n not divisible by the world size or by 4: ragged last shard
    X, y = make_data(n, d)
    Xs = torch.rand(ns, d, generator=torch.Generator().manual_seed(9))

    class GPModel(g.models.ExactGP):
        def __init__(self, x, yy, lik):
            super().__init__(x, yy, lik)
            self.mean_module = g.means.ConstantMean()
            self.covar_module = g.kernels.ScaleKernel(g.kernels.MaternKernel(nu=2.5))

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    if fixed_noise:
        nv = 0.4 + 0.2 * torch.rand(n, generator=torch.Generator().manual_seed(4))
        lik = g.likelihoods.FixedNoiseGaussianLikelihood(noise=nv.to(dev))
    else:
        lik = g.likelihoods.GaussianLikelihood()
    m = GPModel(X.float().to(dev), y.float().to(dev), lik).to(dev)
    m.covar_module.base_kernel.lengthscale = 0.5
    m.covar_module.outputscale = 1.3
    if not fixed_noise:
        lik.noise = 0.5
    m.eval()
    lik.eval()
    S = g.settings
    torch.manual_seed(77)  # the Lanczos start vector of the LOVE cache comes from the default generator (rank 0's, when sharded)
    with torch.no_grad(), S.max_cholesky_size(0), S.eval_cg_tolerance(1e-4), S.fast_pred_var(), S.max_root_decomposition_size(500), \
            S.max_preconditioner_size(precond), S.min_preconditioning_size(100), S.sharding(row_group=row_group), S.rhs_refinement(refine):
        mu = m(Xs.to(dev)).mean  # mean-cache CG (the most recent solve: LOVE below is Lanczos only)
        from gpytorch_amd import linear_cg as LCG

        iters = LCG.LAST_INFO.iterations
        var = m(Xs.to(dev)).variance
    return mu, var, iters


@pytest.mark.parametrize("world,fixed_noise,precond,refine", [(2, False, 0, False), (3, True, 0, False), (2, False, 15, False), (2, True, 15, True)])
def test_row_sharded_posterior_on_device(world, fixed_noise, precond, refine, dev):
    """the small-t solves of the predictive posterior (mean-cache CG, LOVE Lanczos) with every rank owning
    a block of ROWS of K_hat.  `world` processes share cuda:0 (gloo carries the all-gathers of the search directions and the
    per-iteration all-reduces of the solver's partial sums): same CG iteration count, mean and LOVE variance as the
    single-process run, on every rank.  precond = 15: the reference-default pivoted-Cholesky preconditioner, built replicated and
    applied row-sharded (its k x t coefficients all-reduced per apply).  refine: ``settings.rhs_refinement`` on the sharded solve (round 6: the float64
    residual of a rank's rows from one rectangular fused float64 product) against the refined single-process solve."""
    import gpytorch_amd as g

    port = free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_row_worker, args=(r, world, port, q, fixed_noise, precond, refine)) for r in range(world)]
    for p in procs:
        p.start()
    results = _collect(q, procs, world)
    mu, var, iters = _posterior(g, dev, fixed_noise, None, precond, refine)
    mu, var = mu.cpu(), var.cpu()
    assert iters > (3 if precond else 10)
    assert float(var.min()) > 1e-4  # the single-process LOVE variances are themselves converged (nothing clipped)
    for rank, mu_r, var_r, it_r in results:
        mu_r, var_r = torch.from_numpy(mu_r), torch.from_numpy(var_r)
        assert abs(it_r - iters) <= max(4, 0.08 * iters), (rank, it_r, iters)  # float32 summation order, see test_gpu_bbmm
        assert float((mu_r - mu).abs().max()) < 2e-4 * float(mu.abs().max()), rank
        # Lanczos coefficients are chaotic in float32 beyond ~30 steps (any two implementations diverge), so the LOVE
        # caches are compared where the decomposition has converged (noise 0.5: condition number ~4e3, 500 steps)
        assert float((var_r - var).abs().max()) < 1e-2 * float(var.abs().max()), (rank, float((var_r - var).abs().max()), float(var.abs().max()))


def _probe_api_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    import gpytorch_amd as g

    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    n, d = 2500, 3
    X, y = make_data(n, d)

    class GPModel(g.models.ExactGP):
        def __init__(self, x, yy, lik):
            super().__init__(x, yy, lik)
            self.mean_module = g.means.ConstantMean()
            self.covar_module = g.kernels.ScaleKernel(g.kernels.RBFKernel())

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    lik = g.likelihoods.GaussianLikelihood()
    m = GPModel(X.float().to(dev), y.float().to(dev), lik).to(dev)
    m.covar_module.base_kernel.lengthscale = 0.25
    m.covar_module.outputscale = 1.0
    lik.noise = 0.1
    mll = g.ExactMarginalLogLikelihood(lik, m)
    m.train()
    lik.train()
    S = g.settings
    torch.manual_seed(5)  # same seed on every rank: the rank-specific generators must still give different probes
    with S.max_cholesky_size(0), S.num_trace_samples(48), S.max_preconditioner_size(0), S.cg_tolerance(1e-3), \
            S.sharding(probe_group=dist.group.WORLD):
        val = mll(m(m.train_inputs[0]), m.train_targets)
        val.backward()
    grads = [float(p.grad.sum()) for p in (m.covar_module.base_kernel.raw_lengthscale, m.covar_module.raw_outputscale, lik.noise_covar.raw_noise)]
    q.put((rank, float(val), grads))
    dist.barrier()
    dist.destroy_process_group()


def test_probe_sharding_through_the_model_api(dev):
    """settings.sharding(probe_group=...): ExactMarginalLogLikelihood through the gpytorch-shaped API on 2 ranks sharing
    cuda:0; each rank solves 24 of the 48 probes (+ y).  Every rank reports the same value and gradients (all-reduced),
    and they agree with the dense float64 MLL to the accuracy of a 48-probe trace estimate."""
    world, port = 2, free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_probe_api_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = _collect(q, procs, world)
    X, y = make_data(2500, 3)
    ref, gref = OG.dense_mll_and_grads("rbf", X, y, 0.25, 1.0, 0.1)
    (_, v0, g0), (_, v1, g1) = results
    assert abs(v0 - v1) < 1e-6 * max(1.0, abs(v0))
    assert all(abs(a - b) < 1e-5 * max(1e-3, abs(a)) for a, b in zip(g0, g1))
    assert abs(v0 - float(ref)) < 0.02 * max(1.0, abs(float(ref))), (v0, float(ref))
    chain = [1.0 - math.exp(-v) for v in (0.25, 1.0, 0.1 - 1e-4)]
    for gg, rr, cc in zip(g0, gref, chain):
        assert abs(gg - float(rr) * cc) < 0.15 * abs(float(rr) * cc) + 2e-3, (gg, float(rr) * cc)


def test_mll_with_priors_and_lbfgs_training(dev):
    """SURVEY.md 8f rank 1 -- the step around the hot path.  (i) ``test/mlls/test_exact_marginal_log_likelihood.py:71-91``:
    MLL == (log_prob + sum of prior log-probs) / n.  (ii) hyper-parameter training with L-BFGS (strong Wolfe) over the
    deterministic Cholesky branch and then Adam over the BBMM branch, ARD lengthscales, priors on every parameter: the
    loss decreases and the fitted model predicts held-out data (MAE < 0.15, the bar of test_keops_gp_regression.py:58-77)."""
    import gpytorch_amd as g
    from gpytorch_amd import priors as P

    n, d = 700, 2
    X, y = make_data(n + 200, d, seed=3)
    Xt, yt, Xs, ys = X[:n], y[:n], X[n:], y[n:]

    class GPModel(g.models.ExactGP):
        def __init__(self, x, yy, lik):
            super().__init__(x, yy, lik)
            self.mean_module = g.means.ConstantMean(constant_prior=P.NormalPrior(0.0, 1.0))
            base = g.kernels.MaternKernel(nu=2.5, ard_num_dims=d, lengthscale_prior=P.GammaPrior(3.0, 6.0))
            self.covar_module = g.kernels.ScaleKernel(base, outputscale_prior=P.GammaPrior(2.0, 0.15))

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    lik = g.likelihoods.GaussianLikelihood(noise_prior=P.GammaPrior(1.1, 0.05))
    m = GPModel(Xt.float().to(dev), yt.float().to(dev), lik).to(dev)
    mll = g.ExactMarginalLogLikelihood(lik, m)
    S = g.settings
    m.train()
    lik.train()
    # (i) assembly
    out = m(m.train_inputs[0])
    val = mll(out, m.train_targets)
    lp = lik(out).log_prob(m.train_targets)
    prior_sum = sum(pr.log_prob(cl(mod)).sum() for _, mod, pr, cl, _ in m.named_priors())
    assert abs(float(val) - float((lp + prior_sum) / n)) < 1e-5 * max(1.0, abs(float(val)))
    assert len(list(m.named_priors())) == 4
    # (ii) L-BFGS on the exact branch
    opt = torch.optim.LBFGS(m.parameters(), lr=0.5, max_iter=12, line_search_fn="strong_wolfe")
    losses = []

    def closure():
        opt.zero_grad()
        loss = -mll(m(m.train_inputs[0]), m.train_targets)
        loss.backward()
        losses.append(float(loss))
        return loss

    for _ in range(3):
        opt.step(closure)
    assert losses[-1] < losses[0] - 0.05 and all(math.isfinite(v) for v in losses)
    # a few Adam steps on the BBMM branch starting from there (stochastic gradients)
    opt2 = torch.optim.Adam(m.parameters(), lr=0.02)
    with S.max_cholesky_size(0), S.num_trace_samples(16):
        for _ in range(5):
            opt2.zero_grad()
            loss = -mll(m(m.train_inputs[0]), m.train_targets)
            loss.backward()
            opt2.step()
            assert math.isfinite(float(loss))
    m.eval()
    lik.eval()
    with torch.no_grad(), S.max_cholesky_size(0), S.fast_pred_var():
        pred = lik(m(Xs.float().to(dev)))
    assert float((pred.mean.cpu() - ys).abs().mean()) < 0.15
    assert bool((pred.variance > 0).all())


def test_fantasy_model_matches_retrained_posterior(dev):
    """SURVEY.md 8f rank 2 -- ``ExactGP.get_fantasy_model`` (exact_gp.py:151-263; reference test:
    test_simple_gp_regression.py:264-323): conditioning on 40 extra observations through the Schur-complement update
    of the mean cache (one mBCG solve with 40 right-hand sides) gives the same predictive mean as a model holding all
    the data, and as the dense float64 posterior."""
    import gpytorch_amd as g

    n, m, ns, d = 900, 40, 120, 2
    X, y = make_data(n + m + ns, d, seed=11)
    Xt, yt, Xf, yf, Xs = X[:n], y[:n], X[n : n + m], y[n : n + m], X[n + m :]

    class GPModel(g.models.ExactGP):
        def __init__(self, x, yy, lik):
            super().__init__(x, yy, lik)
            self.mean_module = g.means.ConstantMean()
            self.covar_module = g.kernels.ScaleKernel(g.kernels.RBFKernel())

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    def build(xx, yy):
        lik = g.likelihoods.GaussianLikelihood()
        mdl = GPModel(xx.float().to(dev), yy.float().to(dev), lik).to(dev)
        mdl.covar_module.base_kernel.lengthscale = 0.3
        mdl.covar_module.outputscale = 1.2
        lik.noise = 0.05
        mdl.mean_module.constant = 0.1
        mdl.eval()
        lik.eval()
        return mdl

    S = g.settings
    base = build(Xt, yt)
    with pytest.raises(RuntimeError):
        base.get_fantasy_model(Xf.float().to(dev), yf.float().to(dev))
    with torch.no_grad(), S.max_cholesky_size(0), S.eval_cg_tolerance(1e-5), S.skip_posterior_variances():
        base(Xs.float().to(dev))  # builds the caches
        fant = base.get_fantasy_model(Xf.float().to(dev), yf.float().to(dev))
        mu_f = fant(Xs.float().to(dev)).mean
        full = build(torch.cat([Xt, Xf]), torch.cat([yt, yf]))
        mu_full = full(Xs.float().to(dev)).mean
    assert fant.train_inputs[0].shape[0] == n + m and base.train_inputs[0].shape[0] == n
    mu_ref, _ = OG.dense_posterior("rbf", torch.cat([Xt, Xf]), torch.cat([yt, yf]), Xs, 0.3, 1.2, 0.05, mean=0.1)
    assert rel_err(mu_f, mu_full) < 1e-3
    assert rel_err(mu_f, mu_ref) < 1e-3


def test_fantasy_model_updates_love_cache_and_sampling(dev):
    """SURVEY.md 8f rank 2, remainder: (i) ``get_fantasy_model`` UPDATES an existing LOVE covariance cache through the bordered
    inverse (exact_prediction_strategies.py:137-265) -- the fantasy model's ``fast_pred_var`` variances equal the dense
    float64 posterior variances of the enlarged data set within the reference's own LOVE tolerance (5 %,
    test_simple_gp_regression.py:396-442); (ii) ``MultivariateNormal.rsample`` (multivariate_normal.py:254-320) through the
    matrix-free Lanczos root of the predictive covariance (``fast_pred_samples``, settings.py:225-243): sample mean / variance
    match the predictive mean / variance."""
    import gpytorch_amd as g

    n, m, ns, d = 900, 30, 150, 2
    X, y = make_data(n + m + ns, d, seed=12)
    Xt, yt, Xf, yf, Xs = X[:n], y[:n], X[n : n + m], y[n : n + m], X[n + m :]

    class GPModel(g.models.ExactGP):
        def __init__(self, x, yy, lik):
            super().__init__(x, yy, lik)
            self.mean_module = g.means.ConstantMean()
            self.covar_module = g.kernels.ScaleKernel(g.kernels.RBFKernel())

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    lik = g.likelihoods.GaussianLikelihood()
    mdl = GPModel(Xt.float().to(dev), yt.float().to(dev), lik).to(dev)
    mdl.covar_module.base_kernel.lengthscale = 0.3
    mdl.covar_module.outputscale = 1.2
    lik.noise = 0.3
    mdl.eval()
    lik.eval()
    S = g.settings
    torch.manual_seed(5)
    with torch.no_grad(), S.max_cholesky_size(0), S.eval_cg_tolerance(1e-5), S.fast_pred_var(), S.max_root_decomposition_size(400):
        _ = mdl(Xs.float().to(dev)).variance                      # builds mean + LOVE caches
        r_old = mdl.prediction_strategy._covar_cache
        assert r_old is not None
        fant = mdl.get_fantasy_model(Xf.float().to(dev), yf.float().to(dev))
        r_new = fant.prediction_strategy._covar_cache
        assert r_new is not None and r_new.shape == (n + m, r_old.shape[-1] + m)   # updated, not rebuilt
        pred = fant(Xs.float().to(dev))
        mu_f, var_f = pred.mean, pred.variance
    mu_ref, var_ref = OG.dense_posterior("rbf", torch.cat([Xt, Xf]), torch.cat([yt, yf]), Xs, 0.3, 1.2, 0.3, mean=0.0, noise=False)
    assert rel_err(mu_f, mu_ref) < 1e-3
    assert float(((var_f.double().cpu() - var_ref).abs() / var_ref).max()) < 0.05
    # (ii) sampling from the predictive distribution of the original model
    with torch.no_grad(), S.max_cholesky_size(0), S.eval_cg_tolerance(1e-5), S.fast_pred_samples(), S.max_root_decomposition_size(150):
        pred = mdl(Xs.float().to(dev))
        torch.manual_seed(0)
        smp = pred.rsample(torch.Size([4000]))
        assert smp.shape == (4000, ns)
        mu, var = pred.mean, pred.variance
    assert float((smp.mean(0) - mu).abs().max()) < 0.06 * float(var.max().sqrt()) + 0.02
    rel_var = (smp.var(0) - var).abs() / var
    assert float(rel_var.mean()) < 0.06 and float(rel_var.max()) < 0.25


def test_sqrt_inv_matmul_ciq_on_the_fused_operator(dev):
    """``gpytorch.sqrt_inv_matmul`` (gpytorch/__init__.py:252-278) = contour-integral quadrature + msMINRES
    over the fused K*V, against the dense float64 eigendecomposition; and MVN sampling through K^{1/2} eps (``ciq_samples``)."""
    import gpytorch_amd as g

    n, d = 1500, 2
    X, y = make_data(n, d)
    kern = g.kernels.ScaleKernel(g.kernels.RBFKernel()).to(dev)
    kern.base_kernel.lengthscale = 0.3
    kern.outputscale = 1.3
    op = kern(X.float().to(dev)).add_jitter(0.2)
    Kh = 1.3 * OK.rbf(X, X, 0.3, x1_eq_x2=True, direct=True) + 0.2 * torch.eye(n, dtype=torch.float64)
    ev, U = torch.linalg.eigh(Kh)
    rhs = torch.randn(n, 3, generator=torch.Generator().manual_seed(0), dtype=torch.float64)
    with g.settings.max_cholesky_size(0):
        out = g.sqrt_inv_matmul(op, rhs.float().to(dev))
    ref = (U @ torch.diag(ev.rsqrt()) @ U.t()) @ rhs
    assert rel_err(out, ref) < 2e-3
    lhs = torch.randn(5, n, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    res, iq = g.sqrt_inv_matmul(op, rhs.float().to(dev), lhs.float().to(dev))
    assert rel_err(res, lhs @ ref) < 3e-3
    assert rel_err(iq, (lhs @ (U @ torch.diag(1.0 / ev) @ U.t()) * lhs).sum(-1)) < 5e-3
    mvn = g.distributions.MultivariateNormal(torch.zeros(n, device=dev), op)
    with g.settings.ciq_samples():
        torch.manual_seed(0)
        smp = mvn.rsample(torch.Size([3000]))
    emp = (smp.double().cpu().t() @ smp.double().cpu()) / 3000
    assert float((emp.diagonal() - Kh.diagonal()).abs().max() / Kh.diagonal().max()) < 0.15
    assert float((emp - Kh).norm() / Kh.norm()) < 0.12


def test_sqrt_inv_matmul_backward_against_dense_autograd(dev):
    """The backward pass of ``gpytorch.sqrt_inv_matmul`` (gpytorch/__init__.py:252-278; consumer variational/ciq_variational_strategy.py:217):
    gradients of g^T K_hat^{-1/2} b w.r.t. lengthscale, outputscale, noise, the right-hand side and the inputs (ciq.SqrtInvMatmulFn: the
    shifted msMINRES solves re-used by one fused bilinear-derivative pass) against float64 autograd through the eigendecomposition."""
    import gpytorch_amd as g

    n, d = 700, 2
    X, _ = make_data(n, d)
    gen = torch.Generator().manual_seed(3)
    b = torch.randn(n, 2, generator=gen, dtype=torch.float64)
    gvec = torch.randn(n, 2, generator=gen, dtype=torch.float64)
    kern = g.kernels.ScaleKernel(g.kernels.RBFKernel()).to(dev)
    kern.base_kernel.lengthscale = 0.4
    kern.outputscale = 1.3
    noise = torch.tensor([0.3], device=dev, requires_grad=True)
    xd = X.float().to(dev).requires_grad_(True)
    bd = b.float().to(dev).requires_grad_(True)
    with g.settings.max_cholesky_size(0):
        op = kern(xd).add_diagonal(noise)
        out = g.sqrt_inv_matmul(op, bd)
        (out * gvec.float().to(dev)).sum().backward()
    p = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in (0.4, 1.3, 0.3)]
    X64 = X.clone().requires_grad_(True)
    b64 = b.clone().requires_grad_(True)
    Kh = p[1] * OK.rbf(X64, X64, p[0], x1_eq_x2=False, direct=True) + p[2] * torch.eye(n, dtype=torch.float64)
    # float64 reference: autograd through eigh is unstable here (kernel matrices have near-degenerate eigenvalues: 1 / (l_i - l_j) terms), so
    # the square root is taken through the SAME contour-integral identity with a converged rule (40 nodes, exact spectral bounds, dense
    # float64 solves) -- K^{-1/2} = sum_q w_q (K + s_q I)^-1 holds to 1e-12 for any interval containing the spectrum
    from gpytorch_amd.ciq import ciq_weights_shifts

    ev = torch.linalg.eigvalsh(Kh.detach())
    wq, sq = ciq_weights_shifts(float(ev[0]) * 0.9, float(ev[-1]) * 1.1, 40)
    eye = torch.eye(n, dtype=torch.float64)
    ref = sum(w_ * torch.linalg.solve(Kh + s_ * eye, b64) for w_, s_ in zip(wq.tolist(), sq.tolist()))
    U_, = (torch.linalg.eigh(Kh.detach())[1],)
    assert rel_err(ref, (U_ @ torch.diag(ev.rsqrt()) @ U_.t()) @ b) < 1e-9      # the identity itself, against the eigendecomposition (values only)
    assert rel_err(out, ref) < 2e-3
    gref = torch.autograd.grad((ref * gvec).sum(), p + [b64, X64])
    sp = lambda v: 1.0 - math.exp(-v)  # noqa: E731
    got = torch.tensor([float(kern.base_kernel.raw_lengthscale.grad.sum()), float(kern.raw_outputscale.grad.sum()), float(noise.grad.sum())], dtype=torch.float64)
    want = torch.tensor([float(gref[0]) * sp(0.4), float(gref[1]) * sp(1.3), float(gref[2])], dtype=torch.float64)
    assert float((got - want).norm() / want.norm()) < 1e-2, (got, want)
    assert float((bd.grad.double().cpu() - gref[3]).norm() / gref[3].norm()) < 5e-3
    assert float((xd.grad.double().cpu() - gref[4]).norm() / gref[4].norm()) < 2e-2


def test_ciq_square_root_at_size_against_a_tight_cg_solve(dev):
    """K_hat^{-1/2} (K_hat^{-1/2} b) = K_hat^{-1} b at n = 100 000 (the native msMINRES update, 15 shifts x 2 columns x n state): the twice-applied
    contour-integral square root against a tight mBCG solve of the same system."""
    import gpytorch_amd as g
    from gpytorch_amd import backend as B
    from gpytorch_amd.linear_cg import linear_cg

    n, d = 100_000, 3
    X, y = make_data(n, d)
    kern = g.kernels.ScaleKernel(g.kernels.RBFKernel()).to(dev)
    kern.base_kernel.lengthscale = 0.25
    kern.outputscale = 1.0
    Xd = X.float().to(dev)
    rhs = torch.stack([y.float(), torch.randn(n, generator=torch.Generator().manual_seed(2))], -1).to(dev)
    with torch.no_grad(), g.settings.max_cholesky_size(0), g.settings.num_contour_quadrature(20):
        op = kern(Xd).add_jitter(0.5)
        half = g.sqrt_inv_matmul(op, rhs)
        full = g.sqrt_inv_matmul(op, half)
    xp = B.prep_points("rbf", Xd, torch.tensor([0.25]), Xd.mean(0))
    sol_t, info = linear_cg(xp, torch.tensor([1.0], device=dev), torch.tensor([0.5], device=dev), B.to_probe_major(rhs), tolerance=1e-5, max_iter=3000)
    assert info.tolerance_reached
    ref = B.from_probe_major(sol_t, n)
    assert rel_err(full, ref) < 2e-3, rel_err(full, ref)


# ---- round 4: the TWO-DIMENSIONAL split (probe groups x row blocks) through the product's own inv_quad_logdet / linear_cg -----------------
def _grid_groups(dist, P, R, rank):
    """P x R grid, rank = p * R + r: (probe group of this rank = the P ranks with the same r, row group = the R ranks with the same p).
    Every rank creates every group, in the same order (torch.distributed.new_group is collective)."""
    probe_groups = [dist.new_group([p * R + r for p in range(P)]) for r in range(R)]
    row_groups = [dist.new_group([p * R + r for r in range(R)]) for p in range(P)]
    return probe_groups[rank % R], row_groups[rank // R]


def _mll_setup(g, dev, n=2600, precond=0):
    X, y = make_data(n, 3)

    class GPModel(g.models.ExactGP):
        def __init__(self, x, yy, lik):
            super().__init__(x, yy, lik)
            self.mean_module = g.means.ZeroMean()
            self.covar_module = g.kernels.ScaleKernel(g.kernels.RBFKernel())

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    lik = g.likelihoods.GaussianLikelihood()
    m = GPModel(X.float().to(dev), y.float().to(dev), lik).to(dev)
    m.covar_module.base_kernel.lengthscale, m.covar_module.outputscale, lik.noise = 0.25, 1.1, 0.1
    return m, lik


def _mll_and_grads(g, m, lik, precond):
    S = g.settings
    mll = g.ExactMarginalLogLikelihood(lik, m)
    m.train()
    lik.train()
    with S.max_cholesky_size(0), S.cg_tolerance(1e-4), S.num_trace_samples(16), S.max_preconditioner_size(precond), S.min_preconditioning_size(100), \
            S.deterministic_probes(True):
        val = mll(m(m.train_inputs[0]), m.train_targets)
        val.backward()
    from gpytorch_amd import linear_cg as LCG

    grads = [float(p.grad.sum()) for p in (m.covar_module.base_kernel.raw_lengthscale, m.covar_module.raw_outputscale, lik.noise_covar.raw_noise)]
    return float(val.detach()), grads, LCG.LAST_INFO.iterations


def _grid_worker(rank, world, port, q, P, R, precond):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    import gpytorch_amd as g

    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    pg, rg = _grid_groups(dist, P, R, rank)
    torch.manual_seed(5)      # deterministic_probes: every rank draws the same (n, 16) matrix and takes its probe group's columns
    m, lik = _mll_setup(g, dev)
    with g.settings.sharding(probe_group=pg if P > 1 else None, mll_row_group=rg if R > 1 else None):
        val, grads, iters = _mll_and_grads(g, m, lik, precond)
    q.put((rank, val, grads, iters))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("P,R,precond", [(2, 2, 0), (1, 2, 0), (2, 2, 15)])
def test_two_dimensional_split_of_the_mll_on_device(P, R, precond, dev):
    """``settings.sharding(probe_group, mll_row_group)`` -> ``bbmm.inv_quad_logdet_forward(group, row_group)`` -> the PRODUCT's ``linear_cg`` with a
    probe group and a ``RowShard`` at once: P x R processes share cuda:0 (gloo carries the all-gathers of the search directions and the inner
    products over the row group, the 2-float stopping rule and the SLQ sums over the probe group).  Same marginal log likelihood, same
    hyper-parameter gradients and the same iteration count as the single-process evaluation with the same 16 probes, on every rank.
    Replaces ``gpytorch/kernels/multi_device_kernel.py:49-92``; DESIGN section 6."""
    import gpytorch_amd as g

    world, port = P * R, free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grid_worker, args=(r, world, port, q, P, R, precond)) for r in range(world)]
    for p in procs:
        p.start()
    results = _collect(q, procs, world)
    torch.manual_seed(5)
    g.settings.deterministic_probes._drawn.clear()          # (other tests of this process may have drawn an (n, 16) matrix already)
    g.settings.deterministic_probes.probe_vectors = None
    m, lik = _mll_setup(g, dev)
    val, grads, iters = _mll_and_grads(g, m, lik, precond)
    for rank, v, gr, it in results:
        assert abs(it - iters) <= max(2, 0.05 * iters), (rank, it, iters)
        # (row blocks run rectangular launches: other split counts and summation orders than the square single-process product -- float32 mBCG
        # at a 1e-4 residual reproduces the log-det quadrature to a few 1e-4 of its value)
        assert abs(v - val) < 5e-4 * max(1.0, abs(val)), (rank, v, val)
        for a, b in zip(gr, grads):
            assert abs(a - b) < 1e-2 * abs(b) + 1e-6, (rank, gr, grads)
