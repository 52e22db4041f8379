"""GPU: the STRUCTURED operators (sum of kernels, Kronecker multitask, Hadamard multitask) on the shared BBMM forward.

  * the pivoted-Cholesky preconditioner built from kernel ROWS of the summed / Kronecker / Hadamard operator (the reference preconditions
    every ``AddedDiagLinearOperator``: ``gpytorch/settings.py:6-31``, ``kernels/multitask_kernel.py:46-54``, ``kernels/kernel.py:592-632``),
    including the non-constant-diagonal branch (per-task noises, fixed heteroskedastic noise);
  * RQ members inside an AdditiveKernel with two DIFFERENT shape parameters (``kernels/rq_kernel.py:61-74``; C ABI version 2 passes
    alpha explicitly), gradients w.r.t. both alphas;
  * the multitask MLL with its probe columns sharded over two ranks (BASELINE C5: "batched CG over task blocks, 4 x MI355X"; two ranks
    sharing cuda:0 here, gloo carrying the collectives).
Ground truth: dense float64 + torch autograd.
"""
import math
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

from oracle import exact_gp as OG
from oracle import kernels as OK
from oracle import multitask as OM
from tests.util import free_port, make_data, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------------ additive, two RQ members
def _additive_rq_model(g, X, y, dev, dtype=torch.float32):
    class M(g.models.ExactGP):
        def __init__(self, x, yy, lik):
            super().__init__(x, yy, lik)
            self.mean_module = g.means.ZeroMean()
            self.covar_module = g.kernels.ScaleKernel(g.kernels.RQKernel()) + g.kernels.ScaleKernel(g.kernels.RQKernel())

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    lik = g.likelihoods.GaussianLikelihood().to(dev)
    m = M(X.to(dtype).to(dev), y.to(dtype).to(dev), lik).to(dev)
    if dtype == torch.float64:
        m, lik = m.double(), lik.double()
    ka, kb = m.covar_module.kernels
    ka.base_kernel.lengthscale, ka.base_kernel.alpha, ka.outputscale = 0.25, 0.7, 1.1
    kb.base_kernel.lengthscale, kb.base_kernel.alpha, kb.outputscale = 0.9, 2.5, 0.4
    lik.noise = 0.1
    return m, lik


@pytest.mark.parametrize("precond,dtype", [(0, torch.float32), (30, torch.float32), (30, torch.float64)], ids=["f32-0", "f32-30", "f64-30"])
def test_additive_of_two_rq_kernels_with_different_alpha(precond, dtype, dev):
    """ScaleKernel(RQ, alpha = 0.7) + ScaleKernel(RQ, alpha = 2.5) on the BBMM path: value and EVERY gradient (both lengthscales, both
    alphas, both outputscales, the noise) against dense float64 autograd; with and without the row-built preconditioner.  float64 (round 6): the
    members of a STRUCTURED operator on the generic path -- fused float64 products, the shape-parameter sums from the row-block derivative."""
    import gpytorch_amd as g

    n, d = 1800, 2
    X, y = make_data(n, d)
    m, lik = _additive_rq_model(g, X, y, dev, dtype)
    ka, kb = m.covar_module.kernels
    mll = g.ExactMarginalLogLikelihood(lik, m)
    m.train()
    lik.train()
    S = g.settings
    with S.max_cholesky_size(0), S.cg_tolerance(1e-5), S.num_trace_samples(400), S.max_preconditioner_size(precond), S.min_preconditioning_size(100):
        torch.manual_seed(0)
        val = mll(m(m.train_inputs[0]), m.train_targets)
        val.backward()
    vals = (0.25, 0.7, 1.1, 0.9, 2.5, 0.4, 0.1)
    p = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in vals]
    Kh = (p[2] * OK.rq(X, X, p[0], p[1], x1_eq_x2=True, direct=True) + p[5] * OK.rq(X, X, p[3], p[4], x1_eq_x2=True, direct=True)
          + p[6] * torch.eye(n, dtype=torch.float64))
    ref = OG.dense_log_prob(Kh, y) / n
    gref = torch.autograd.grad(ref, p)
    assert abs(float(val) - float(ref)) < 5e-3 * max(1.0, abs(float(ref)))
    sp = lambda v: 1.0 - math.exp(-v)  # noqa: E731
    got = [ka.base_kernel.raw_lengthscale.grad, ka.base_kernel.raw_alpha.grad, ka.raw_outputscale.grad,
           kb.base_kernel.raw_lengthscale.grad, kb.base_kernel.raw_alpha.grad, kb.raw_outputscale.grad, lik.noise_covar.raw_noise.grad]
    assert all(gg is not None for gg in got)
    chain = [sp(v) for v in vals[:6]] + [sp(0.1 - 1e-4)]
    gv = torch.tensor([float(gg.sum()) for gg in got], dtype=torch.float64)
    wv = torch.tensor([float(rr) * cc for rr, cc in zip(gref, chain)], dtype=torch.float64)
    assert float((gv - wv).norm() / wv.norm()) < 0.12, (gv, wv)
    # the two shape-parameter gradients individually have the right sign and size when they are not tiny
    for k_ in (1, 4):
        if abs(float(wv[k_])) > 0.05 * float(wv.abs().max()):
            assert abs(float(gv[k_]) - float(wv[k_])) < 0.3 * abs(float(wv[k_])), (k_, gv, wv)


# ------------------------------------------------------------------------------------------------ preconditioners
def _iterations():
    from gpytorch_amd import linear_cg as LCG

    return LCG.LAST_INFO.iterations


def test_sum_operator_preconditioner(dev):
    """(RBF + Matern-5/2) + noise: the row-built preconditioner leaves the solution unchanged, cuts the iteration count, and the
    preconditioned MLL (probes from N(0, P), log|P| correction) lands on the dense value."""
    import gpytorch_amd as g

    n, d = 2600, 2
    X, y = make_data(n, d)
    S = g.settings

    def build():
        kern = (g.kernels.ScaleKernel(g.kernels.RBFKernel()) + g.kernels.ScaleKernel(g.kernels.MaternKernel(nu=2.5))).to(dev)
        ka, kb = kern.kernels
        ka.base_kernel.lengthscale, ka.outputscale = 0.25, 1.1
        kb.base_kernel.lengthscale, kb.outputscale = 0.9, 0.4
        return kern(X.float().to(dev)).add_diagonal(torch.tensor(0.05, device=dev))

    Kh = (1.1 * OK.rbf(X, X, 0.25, x1_eq_x2=True, direct=True) + 0.4 * OK.matern(X, X, 0.9, 2.5, x1_eq_x2=True, direct=True)
          + 0.05 * torch.eye(n, dtype=torch.float64))
    ref = torch.linalg.solve(Kh, y.unsqueeze(-1))
    its, sols = {}, {}
    for rank in (0, 50):
        op = build()
        with torch.no_grad(), S.max_cholesky_size(0), S.cg_tolerance(1e-4), S.max_preconditioner_size(rank), S.min_preconditioning_size(100):
            sols[rank] = op.solve(y.float().to(dev).unsqueeze(-1))
        its[rank] = _iterations()
        assert rel_err(sols[rank], ref) < 2e-3, rank
    assert its[50] < 0.6 * its[0], its
    # MLL ingredients with the preconditioner on
    op = build()
    torch.manual_seed(1)
    with S.max_cholesky_size(0), S.cg_tolerance(1e-4), S.max_preconditioner_size(50), S.min_preconditioning_size(100), S.num_trace_samples(128), \
            S.max_lanczos_quadrature_iterations(60):
        iq, ld = op.inv_quad_logdet(y.float().to(dev).unsqueeze(-1), logdet=True)
    assert abs(float(iq) - float(y @ ref.squeeze(-1))) < 1e-3 * abs(float(y @ ref.squeeze(-1)))
    ld_ref = float(torch.logdet(Kh))
    assert abs(float(ld) - ld_ref) < 0.01 * abs(ld_ref), (float(ld), ld_ref)


T = 3
MT = dict(Bf=torch.tensor([[0.9], [-0.5], [0.7]], dtype=torch.float64), v=torch.tensor([0.4, 0.6, 0.3], dtype=torch.float64),
          tn=torch.tensor([0.05, 0.1, 0.08], dtype=torch.float64))


def _mt_data(n, d, seed=0):
    gg = torch.Generator().manual_seed(seed)
    X = torch.rand(n, d, generator=gg, dtype=torch.float64)
    Y = torch.stack([torch.sin(2 * math.pi * X[:, 0]) + 0.3 * X[:, 1], torch.cos(2 * math.pi * X[:, 0]), torch.sin(math.pi * X.sum(-1))], -1)
    return X, Y + 0.1 * torch.randn(n, T, generator=gg, dtype=torch.float64)


def _mt_model(g, X, Y, dev, ls=0.35):
    class MTM(g.models.ExactGP):
        def __init__(self, x, y, lik):
            super().__init__(x, y, lik)
            self.mean_module = g.means.MultitaskMean(g.means.ZeroMean(), num_tasks=T)
            self.covar_module = g.kernels.MultitaskKernel(g.kernels.RBFKernel(), num_tasks=T, rank=1)

        def forward(self, x):
            return g.distributions.MultitaskMultivariateNormal(self.mean_module(x), self.covar_module(x))

    lik = g.likelihoods.MultitaskGaussianLikelihood(num_tasks=T, has_global_noise=False).to(dev)
    m = MTM(X.float().to(dev), Y.float().to(dev), lik).to(dev)
    m.covar_module.data_covar_module.lengthscale = ls
    with torch.no_grad():
        m.covar_module.task_covar_module.covar_factor.copy_(MT["Bf"].float())
    m.covar_module.task_covar_module.var = MT["v"].float()
    lik.task_noises = MT["tn"].float()
    return m, lik


def test_kronecker_operator_preconditioner_with_per_task_noise(dev):
    """K_XX (x) K_TT + I (x) diag(task noises): rows of the Kronecker product feed the pivoted Cholesky, the three DIFFERENT task noises
    take the non-constant-diagonal branch (P = L L^T + D).  Same solution, fewer iterations, MLL ingredients on the dense values."""
    import gpytorch_amd as g

    n = 900
    X, Y = _mt_data(n, 2)
    S = g.settings
    Kh = OM.khat("rbf", X, 0.35, 1.0, MT["Bf"], MT["v"], MT["tn"])
    yv = Y.reshape(-1)
    ref = torch.linalg.solve(Kh, yv.unsqueeze(-1))
    its = {}
    for rank in (0, 60):
        m, lik = _mt_model(g, X, Y, dev)
        op = lik(m.train()(m.train_inputs[0])).lazy_covariance_matrix
        with torch.no_grad(), S.max_cholesky_size(0), S.cg_tolerance(1e-4), S.max_preconditioner_size(rank), S.min_preconditioning_size(100):
            sol = op.solve(yv.float().to(dev).unsqueeze(-1))
        its[rank] = _iterations()
        assert rel_err(sol, ref) < 2e-3, rank
    assert its[60] < 0.7 * its[0], its
    m, lik = _mt_model(g, X, Y, dev)
    op = lik(m.train()(m.train_inputs[0])).lazy_covariance_matrix
    torch.manual_seed(2)
    with S.max_cholesky_size(0), S.cg_tolerance(1e-4), S.max_preconditioner_size(60), S.min_preconditioning_size(100), S.num_trace_samples(128), \
            S.max_lanczos_quadrature_iterations(60):
        iq, ld = op.inv_quad_logdet(yv.float().to(dev).unsqueeze(-1), logdet=True)
    iq_ref = float(yv @ ref.squeeze(-1))
    assert abs(float(iq) - iq_ref) < 1e-3 * abs(iq_ref)
    ld_ref = float(torch.logdet(Kh))
    assert abs(float(ld) - ld_ref) < 0.01 * abs(ld_ref), (float(ld), ld_ref)


def test_hadamard_operator_preconditioner(dev):
    import gpytorch_amd as g

    n = 2400
    gen = torch.Generator().manual_seed(0)
    x = torch.rand(n, 2, generator=gen)
    i = torch.randint(0, 2, (n,), generator=gen)
    y = torch.where(i == 0, torch.sin(4 * x[:, 0]), torch.cos(3 * x[:, 1])) + 0.1 * torch.randn(n, generator=gen)
    Bf, v = torch.tensor([[0.9], [-0.4]]), torch.tensor([0.3, 0.5])
    ktt = (Bf @ Bf.t() + torch.diag(v)).double()
    Kh = OK.rbf(x.double(), x.double(), 0.35, x1_eq_x2=True, direct=True) * ktt[i][:, i] + 0.05 * torch.eye(n, dtype=torch.float64)
    ref = torch.linalg.solve(Kh, y.double().unsqueeze(-1))
    S = g.settings
    its = {}
    for rank in (0, 50):
        kx = g.kernels.RBFKernel().to(dev)
        kx.lengthscale = 0.35
        kt = g.kernels.IndexKernel(num_tasks=2, rank=1).to(dev)
        with torch.no_grad():
            kt.covar_factor.copy_(Bf)
        kt.var = v
        op = kx(x.to(dev)).mul(kt(i.to(dev))).add_diagonal(torch.tensor(0.05, device=dev))
        with torch.no_grad(), S.max_cholesky_size(0), S.cg_tolerance(1e-4), S.max_preconditioner_size(rank), S.min_preconditioning_size(100):
            sol = op.solve(y.to(dev).unsqueeze(-1))
        its[rank] = _iterations()
        assert rel_err(sol, ref) < 2e-3, rank
    assert its[50] < 0.7 * its[0], its


def test_fixed_noise_operator_is_preconditioned(dev):
    """FixedNoiseGaussianLikelihood (heteroskedastic diagonal): the single-kernel operator now takes the non-constant-diagonal
    preconditioner instead of none."""
    import gpytorch_amd as g
    from gpytorch_amd.operators import FusedKernelAddedDiagLinearOperator

    n = 2600
    X, y = make_data(n, 3)
    noise = 0.02 + 0.1 * torch.rand(n, generator=torch.Generator().manual_seed(9), dtype=torch.float64)
    Kh = OK.kernel_matrix("rbf", X, X, 0.3, 1.1, x1_eq_x2=True) + torch.diag(noise)
    ref = torch.linalg.solve(Kh, y.unsqueeze(-1))
    S = g.settings
    its = {}
    for rank in (0, 50):
        kern = g.kernels.ScaleKernel(g.kernels.RBFKernel()).to(dev)
        kern.base_kernel.lengthscale, kern.outputscale = 0.3, 1.1
        op = kern(X.float().to(dev)).evaluate_kernel() + g.operators.DiagLinearOperator(noise.float().to(dev))
        assert isinstance(op, FusedKernelAddedDiagLinearOperator) and op.noise_vec is not None
        with torch.no_grad(), S.max_cholesky_size(0), S.cg_tolerance(1e-4), S.max_preconditioner_size(rank), S.min_preconditioning_size(100):
            sol = op.solve(y.float().to(dev).unsqueeze(-1))
        its[rank] = _iterations()
        assert rel_err(sol, ref) < 2e-3, rank
    assert its[50] < 0.7 * its[0], its


# ------------------------------------------------------------------------------------------------ C5: sharded multitask MLL
def _mt_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    import gpytorch_amd as g

    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    X, Y = _mt_data(700, 2)
    m, lik = _mt_model(g, X, Y, dev)
    mll = g.ExactMarginalLogLikelihood(lik, m)
    m.train()
    lik.train()
    S = g.settings
    torch.manual_seed(5)  # same seed everywhere: the rank-specific generators must still give different probes
    with S.max_cholesky_size(0), S.num_trace_samples(64), S.max_preconditioner_size(20), S.min_preconditioning_size(100), S.cg_tolerance(1e-3), \
            S.sharding(probe_group=dist.group.WORLD):
        val = mll(m(m.train_inputs[0]), m.train_targets)
        val.backward()
    from gpytorch_amd import linear_cg as LCG

    grads = [m.covar_module.data_covar_module.raw_lengthscale.grad.reshape(-1), m.covar_module.task_covar_module.covar_factor.grad.reshape(-1),
             m.covar_module.task_covar_module.raw_var.grad.reshape(-1), lik.raw_task_noises.grad.reshape(-1)]
    q.put((rank, float(val), torch.cat(grads).double().cpu().tolist(), LCG.LAST_INFO.residual_norms.numel()))
    dist.barrier()
    dist.destroy_process_group()


def test_multitask_mll_probe_sharded_over_two_ranks(dev):
    """BASELINE C5's split in miniature: the Kronecker MLL with 64 probes sharded 32 + 32 over two ranks (y on rank 0).  Both ranks
    report the same value and (all-reduced) gradients; they agree with the dense float64 MLL to the accuracy of a 64-probe estimate."""
    world, port = 2, free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_mt_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time

    results, t0 = [], time.time()
    while len(results) < world:
        try:
            results.append(q.get(timeout=2.0))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > 300:
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                raise AssertionError(f"rank process failed (exit codes {dead}) or timed out")
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, v0, g0, c0), (_, v1, g1, c1) = sorted(results)
    assert {c0, c1} == {32, 33}                       # columns in the last solve: 32 probes (+ y on rank 0)
    assert abs(v0 - v1) < 1e-6 * max(1.0, abs(v0))
    assert all(abs(a - b) < 1e-5 * max(1e-3, abs(a)) for a, b in zip(g0, g1))
    X, Y = _mt_data(700, 2)
    ref, gref = OM.dense_mll_and_grads("rbf", X, Y, 0.35, 1.0, MT["Bf"], MT["v"], MT["tn"])
    assert abs(v0 - float(ref)) < 0.02 * max(1.0, abs(float(ref))), (v0, float(ref))
    sg = lambda a: 1.0 - torch.exp(-a)  # noqa: E731
    want = torch.cat([(gref[0] * (1 - math.exp(-0.35))).reshape(-1), gref[1].reshape(-1), (gref[2] * sg(MT["v"])).reshape(-1),
                      (gref[3] * sg(MT["tn"] - 1e-4)).reshape(-1)])
    got = torch.tensor(g0, dtype=torch.float64)
    assert float((got - want).norm() / want.norm()) < 0.15, (got, want)


# ------------------------------------------------------------------------------------------------ RQ members in float64 (round 6)
@pytest.mark.parametrize("structure", ["kronecker", "hadamard"])
def test_rq_data_kernel_in_float64_on_the_structured_operators(structure, dev):
    """Round 5 left "RQ members of the structured operators in float64" open.  A float64 model with an RQ data kernel under the Kronecker multitask
    operator and under the Hadamard (observed-task) operator, BBMM branch with a COMPLETE probe basis (exact trace): value and the gradients with
    respect to lengthscale and alpha against dense float64 autograd."""
    import gpytorch_amd as g

    S = g.settings
    ls, alpha = 0.4, 1.3
    gen = torch.Generator().manual_seed(3)
    if structure == "kronecker":
        n, T = 120, 2
        X = torch.rand(n, 2, generator=gen, dtype=torch.float64)
        Y = torch.stack([torch.sin(4 * X[:, 0]), torch.cos(3 * X[:, 1])], -1) + 0.1 * torch.randn(n, T, generator=gen, dtype=torch.float64)
        Bf, v, tn = torch.tensor([[0.8], [-0.5]], dtype=torch.float64), torch.tensor([0.4, 0.6], dtype=torch.float64), torch.tensor([0.05, 0.1], dtype=torch.float64)

        class MT(g.models.ExactGP):
            def __init__(self, x, y, lik):
                super().__init__(x, y, lik)
                self.mean_module = g.means.MultitaskMean(g.means.ZeroMean(), num_tasks=T)
                self.covar_module = g.kernels.MultitaskKernel(g.kernels.RQKernel(), num_tasks=T, rank=1)

            def forward(self, x):
                return g.distributions.MultitaskMultivariateNormal(self.mean_module(x), self.covar_module(x))

        lik = g.likelihoods.MultitaskGaussianLikelihood(num_tasks=T, has_global_noise=False).to(dev).double()
        m = MT(X.to(dev), Y.to(dev), lik).to(dev).double()
        kx = m.covar_module.data_covar_module
        kx.lengthscale, kx.alpha = ls, alpha
        with torch.no_grad():
            m.covar_module.task_covar_module.covar_factor.copy_(Bf)
        m.covar_module.task_covar_module.var = v
        lik.task_noises = tn
        N = n * T
        args, target = (m.train_inputs[0],), m.train_targets

        def dense(p_ls, p_a):
            Kx = OK.rq(X, X, p_ls, p_a, x1_eq_x2=True, direct=True)
            return torch.kron(Kx, Bf @ Bf.t() + torch.diag(v)) + torch.diag(tn.repeat(n)), Y.reshape(-1)
    else:
        n = 300
        X = torch.rand(n, 2, generator=gen, dtype=torch.float64)
        idx = torch.randint(0, 2, (n,), generator=gen)
        yv = torch.where(idx == 0, torch.sin(4 * X[:, 0]), torch.cos(3 * X[:, 1])) + 0.1 * torch.randn(n, generator=gen, dtype=torch.float64)
        Bf, v = torch.tensor([[0.9], [-0.4]], dtype=torch.float64), torch.tensor([0.3, 0.5], dtype=torch.float64)

        class HM(g.models.ExactGP):
            def __init__(self, train_x, train_y, likelihood):
                super().__init__(train_x, train_y, likelihood)
                self.mean_module = g.means.ZeroMean()
                self.covar_module = g.kernels.RQKernel()
                self.task_covar_module = g.kernels.IndexKernel(num_tasks=2, rank=1)

            def forward(self, x, i):
                return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x).mul(self.task_covar_module(i)))

        lik = g.likelihoods.GaussianLikelihood().to(dev).double()
        m = HM((X.to(dev), idx.to(dev)), yv.to(dev), lik).to(dev).double()
        kx = m.covar_module
        kx.lengthscale, kx.alpha = ls, alpha
        with torch.no_grad():
            m.task_covar_module.covar_factor.copy_(Bf)
        m.task_covar_module.var = v
        lik.noise = 0.1
        N = n
        args, target = m.train_inputs, m.train_targets

        def dense(p_ls, p_a):
            Kx = OK.rq(X, X, p_ls, p_a, x1_eq_x2=True, direct=True)
            return Kx * (Bf @ Bf.t() + torch.diag(v))[idx][:, idx] + 0.1 * torch.eye(n, dtype=torch.float64), yv

    mll = g.ExactMarginalLogLikelihood(lik, m)
    m.train(), lik.train()
    S.deterministic_probes.probe_vectors = math.sqrt(N) * torch.eye(N, dtype=torch.float64)
    try:
        with S.max_cholesky_size(0), S.cg_tolerance(1e-6), S.deterministic_probes(True), S.max_preconditioner_size(0), S.max_lanczos_quadrature_iterations(80):
            val = mll(m(*args), target)
            val.backward()
    finally:
        S.deterministic_probes.probe_vectors = None
    p_ls = torch.tensor(ls, dtype=torch.float64, requires_grad=True)
    p_a = torch.tensor(alpha, dtype=torch.float64, requires_grad=True)
    Kh, yy = dense(p_ls, p_a)
    ref = OG.dense_log_prob(Kh, yy) / N
    g_ls, g_a = torch.autograd.grad(ref, [p_ls, p_a])
    assert val.dtype == torch.float64
    assert abs(float(val) - float(ref)) < 5e-3 * max(1.0, abs(float(ref))), (float(val), float(ref))     # (SLQ value on an exact probe basis: Lanczos steps)
    sp = lambda t_: 1.0 - math.exp(-t_)  # noqa: E731
    got_ls, got_a = float(kx.raw_lengthscale.grad.sum()), float(kx.raw_alpha.grad.sum())
    assert abs(got_ls - float(g_ls) * sp(ls)) < 2e-2 * abs(float(g_ls) * sp(ls)) + 1e-5, (got_ls, float(g_ls) * sp(ls))
    assert abs(got_a - float(g_a) * sp(alpha)) < 2e-2 * abs(float(g_a) * sp(alpha)) + 1e-5, (got_a, float(g_a) * sp(alpha))
