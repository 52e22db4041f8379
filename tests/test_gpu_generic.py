"""GPU parity of the GENERIC path: float64 models and inputs with more than 32 dimensions.

The reference honours the dtype of its inputs (its own model tests run in float32 and float64; gradcheck-style tests are
float64) and has no limit on the input dimension.  Here float32 / d <= 32 goes through the fused MFMA kernels and
everything else through HIP-generated row blocks of K x library GEMM + the float64 instantiation of the device-resident
mBCG kernels.  Ground truth: the float64 oracle (dense Cholesky and the restated ``linear_cg``).

Tolerances: float64 kernel entries 1e-12; float64 CG is compared ITERATION FOR ITERATION with the float64 oracle
(same counts, solves to 1e-5, early Lanczos coefficients to 1e-8); MLL / gradients / posterior at the rtol 1e-3 of BASELINE.json's north_star.
"""
import math

import pytest
import torch

from oracle import exact_gp as OG
from oracle import kernels as OK
from oracle.linear_cg import linear_cg as oracle_cg
from tests.util import make_data, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,d", [("rbf", 3), ("matern12", 2), ("matern32", 7), ("matern52", 20)])
def test_float64_kernel_entries_and_product(kind, d, dev):
    from gpytorch_amd import backend as B

    n, m, t = 700, 433, 9
    g0 = torch.Generator().manual_seed(5)
    X1 = torch.rand(n, d, generator=g0, dtype=torch.float64)
    X2 = torch.rand(m, d, generator=g0, dtype=torch.float64)
    V = torch.randn(m, t, generator=g0, dtype=torch.float64)
    ls = 0.3 + 0.4 * torch.rand(d, generator=g0, dtype=torch.float64)
    K = OK.kernel_matrix(kind, X1, X2, ls.reshape(1, d), 1.7, x1_eq_x2=False, direct=True)
    shift = X1.mean(0).to(dev)
    p1 = B.prep_points(kind, X1.to(dev), ls.to(dev), shift)
    p2 = B.prep_points(kind, X2.to(dev), ls.to(dev), shift)
    assert p1.dtype == torch.float64 and not p1.fused
    sc = torch.tensor([1.7], dtype=torch.float64, device=dev)
    assert rel_err(B.kernel_dense(p1, p2, sc), K) < 1e-12
    rows = torch.tensor([0, 5, n - 1, 17])
    assert rel_err(B.kernel_rows(p1, rows, p2, sc), K[rows]) < 1e-12
    assert rel_err(B.kernel_diag(p1, p1, sc), torch.full((n,), 1.7, dtype=torch.float64)) < 1e-12
    out = B.from_probe_major(B.kv(p1, p2, B.to_probe_major(V.to(dev), torch.float64), scale=sc), n)
    assert out.dtype == torch.float64
    assert rel_err(out, K @ V) < 1e-12


def test_float64_cg_matches_oracle_iteration_for_iteration(dev):
    """Device-resident mBCG instantiated for double vs the restated float64 ``linear_cg``: same iteration count (+-1),
    same early tridiagonal rows (1e-8; late rows are chaotic in any finite precision), same quadrature, solves to the
    CG tolerance; then the rank-15-preconditioned float64 solve is checked through its true residual."""
    from gpytorch_amd import backend as B
    from gpytorch_amd.bbmm import build_preconditioner
    from gpytorch_amd.linear_cg import linear_cg

    kind, n, d, ls, s2, t = "rbf", 1500, 2, 0.3, 0.05, 6
    X, y = make_data(n, d)
    g0 = torch.Generator().manual_seed(2)
    rhs = torch.cat([torch.randn(n, t - 1, generator=g0, dtype=torch.float64), y.unsqueeze(-1)], -1)
    mm = OG.make_matmul(kind, X, ls, 1.0, s2)
    ref, tm, oinfo = oracle_cg(mm, rhs, n_tridiag=t - 1, tolerance=1e-4, max_iter=400, max_tridiag_iter=30, return_info=True)
    xp = B.prep_points(kind, X.to(dev), torch.tensor(ls, dtype=torch.float64))
    sc = torch.ones(1, dtype=torch.float64, device=dev)
    nz = torch.full((1,), s2, dtype=torch.float64, device=dev)
    sol_t, info = linear_cg(xp, sc, nz, B.to_probe_major(rhs.to(dev), torch.float64), n_tridiag=t - 1, tolerance=1e-4,
                            max_iter=400, max_tridiag_iter=30)
    assert sol_t.dtype == torch.float64
    sol = B.from_probe_major(sol_t, n)
    # (the eps = 1e-10 masks of the algorithm stall any CG near 1e-5 relative residual -- in the oracle too -- so the
    # comparison runs at tolerance 1e-4, where both stop by the rule and on the same iteration)
    assert info.tolerance_reached and oinfo["tolerance_reached"]
    assert info.iterations == oinfo["iters"]
    assert rel_err(sol, ref) < 1e-5
    assert info.t_mats.shape == tm.shape
    assert rel_err(info.t_mats[:, :8, :8], tm[:, :8, :8]) < 1e-8
    from oracle import slq as OS
    from gpytorch_amd.bbmm import slq_logdet
    assert abs(float(slq_logdet(info.t_mats, n)) - float(OS.slq_logdet(tm, n))) < 1e-6 * abs(float(OS.slq_logdet(tm, n)))
    # preconditioned float64 solve: true residual below the tolerance
    pre = build_preconditioner(xp, sc, nz, rank=15, min_size=0)
    assert pre is not None and pre.q1t.dtype == torch.float64
    sol_t, info = linear_cg(xp, sc, nz, B.to_probe_major(rhs.to(dev), torch.float64), tolerance=1e-4, max_iter=400, preconditioner=pre)
    assert info.tolerance_reached
    sol = B.from_probe_major(sol_t, n).cpu()
    res = (mm(sol) - rhs).norm(dim=0) / rhs.norm(dim=0)
    assert float(res.mean()) < 1.2e-4


def _model(kind, X, y, ls, os_, noise, dev, dtype, ard=False, mean=0.0):
    import gpytorch_amd as g

    class GPModel(g.models.ExactGP):
        def __init__(self, x, yy, lik):
            super().__init__(x, yy, lik)
            self.mean_module = g.means.ConstantMean()
            d = x.shape[-1]
            if kind == "rbf":
                base = g.kernels.RBFKernel(ard_num_dims=d if ard else None)
            else:
                base = g.kernels.MaternKernel(nu=OK.KINDS[kind], ard_num_dims=d if ard else None)
            self.covar_module = g.kernels.ScaleKernel(base)

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    lik = g.likelihoods.GaussianLikelihood()
    m = GPModel(X.to(dev, dtype), y.to(dev, dtype), lik).to(dev).to(dtype)
    m.covar_module.base_kernel.lengthscale = ls
    m.covar_module.outputscale = os_
    lik.noise = noise
    m.mean_module.constant = mean
    return g, m, lik


def _chain(*vals):
    return [1.0 - math.exp(-v) for v in vals]


@pytest.mark.parametrize("kind,d,dtype", [("rbf", 3, torch.float64), ("matern52", 40, torch.float32), ("matern32", 20, torch.float64)])
def test_mll_and_grads_generic_path(kind, d, dtype, dev):
    """BBMM branch (max_cholesky_size 0) with a COMPLETE probe basis sqrt(n) I, so the trace terms are exact and the
    comparison with the dense float64 MLL is deterministic; both the Cholesky branch and the BBMM branch are checked."""
    n, ls, os_, s2 = 260, 0.9 if d > 10 else 0.3, 1.4, 0.1
    X, y = make_data(n, d)
    ref, gref = OG.dense_mll_and_grads(kind, X, y, ls, os_, s2, mean=0.1)
    c = _chain(ls, os_, s2 - 1e-4)
    for bbmm in (False, True):
        g, m, lik = _model(kind, X, y, ls, os_, s2, dev, dtype, mean=0.1)
        mll = g.ExactMarginalLogLikelihood(lik, m)
        m.train()
        lik.train()
        if bbmm:
            Z = math.sqrt(n) * torch.eye(n, dtype=torch.float64)
            with g.settings.max_cholesky_size(0), g.settings.deterministic_probes(True), g.settings.cg_tolerance(1e-5), \
                    g.settings.max_preconditioner_size(0), g.settings.max_lanczos_quadrature_iterations(n):
                g.settings.deterministic_probes.probe_vectors = Z.to(dev)
                try:
                    val = mll(m(m.train_inputs[0]), m.train_targets)
                    val.backward()
                finally:
                    g.settings.deterministic_probes.probe_vectors = None
        else:
            val = mll(m(m.train_inputs[0]), m.train_targets)
            val.backward()
        assert val.dtype == dtype
        tol = 2e-3 if bbmm else 1e-3
        assert abs(float(val) - float(ref)) < tol * max(1.0, abs(float(ref))), (bbmm, float(val), float(ref))
        got = (m.covar_module.base_kernel.raw_lengthscale.grad, m.covar_module.raw_outputscale.grad, lik.noise_covar.raw_noise.grad)
        for gg, rr, cc in zip(got, gref, c):
            assert abs(float(gg.sum()) - float(rr) * cc) < 5 * tol * abs(float(rr) * cc) + 2e-5, (kind, bbmm, float(gg.sum()), float(rr) * cc)


def test_generic_gradient_kernel_vs_fused_and_autograd(dev):
    """``kv_grad_generic`` (row blocks + gpamd_kernel_grad_block) against float64 autograd, ARD, rectangular, d = 35 (beyond the fused kernels' 32)."""
    from gpytorch_amd import backend as B
    from gpytorch_amd.functions import hyper_grads

    n, m_, d, t = 333, 517, 35, 11
    g0 = torch.Generator().manual_seed(4)
    X1 = torch.rand(n, d, generator=g0, dtype=torch.float64)
    X2 = torch.rand(m_, d, generator=g0, dtype=torch.float64)
    Lm = torch.randn(n, t, generator=g0, dtype=torch.float64)
    Rm = torch.randn(m_, t, generator=g0, dtype=torch.float64)
    ls = (0.8 + 0.4 * torch.rand(1, d, generator=g0, dtype=torch.float64)).requires_grad_(True)
    os_ = torch.tensor(1.7, dtype=torch.float64, requires_grad=True)
    for kind in ("rbf", "matern52", "matern32", "matern12"):
        K = OK.kernel_matrix(kind, X1, X2, ls, os_, x1_eq_x2=False, direct=True)
        gl, go = torch.autograd.grad((Lm * (K @ Rm)).sum(), [ls, os_])
        for dtype, tol in ((torch.float64, 1e-9), (torch.float32, 1e-3)):
            shift = X1.mean(0).to(dev, dtype)
            lsd = ls.detach().to(dev, dtype)
            p1 = B.prep_points(kind, X1.to(dev, dtype), lsd, shift)
            p2 = B.prep_points(kind, X2.to(dev, dtype), lsd, shift)
            assert not p1.fused
            d_ls, d_os = hyper_grads(p1, p2, lsd, os_.detach().to(dev, dtype).reshape(1), B.to_probe_major(Lm.to(dev), dtype),
                                     B.to_probe_major(Rm.to(dev), dtype))
            assert rel_err(d_ls, gl) < tol, (kind, dtype)
            assert abs(float(d_os) - float(go)) < tol * abs(float(go)), (kind, dtype)


def test_generic_path_equals_fused_path_on_the_same_problem(dev):
    """Same float32 problem down both routes (FORCE_GENERIC): K V and the whole mBCG solve agree to float32 rounding."""
    from gpytorch_amd import backend as B
    from gpytorch_amd.linear_cg import linear_cg

    kind, n, d, ls, s2, t = "matern52", 3000, 4, 0.5, 0.05, 12
    X, y = make_data(n, d)
    g0 = torch.Generator().manual_seed(3)
    rhs = torch.cat([torch.randn(n, t - 1, generator=g0), y.float().unsqueeze(-1)], -1)
    xp = B.prep_points(kind, X.float().to(dev), torch.tensor(ls), X.mean(0).float().to(dev))
    sc = torch.ones(1, device=dev)
    nz = torch.full((1,), s2, device=dev)
    vt = B.to_probe_major(rhs.to(dev))
    fused = B.kv(xp, xp, vt, scale=sc, dscale=nz, vd=vt)
    sol_f, info_f = linear_cg(xp, sc, nz, vt, n_tridiag=4, tolerance=1e-3, max_iter=300)
    B.FORCE_GENERIC = True
    try:
        assert not xp.fused
        gen = B.kv(xp, xp, vt, scale=sc, dscale=nz, vd=vt)
        sol_g, info_g = linear_cg(xp, sc, nz, vt, n_tridiag=4, tolerance=1e-3, max_iter=300)
    finally:
        B.FORCE_GENERIC = False
    assert rel_err(gen, fused) < 2e-5
    assert abs(info_f.iterations - info_g.iterations) <= 6  # +-3 % of ~200: float32 summation order, see test_gpu_bbmm
    assert rel_err(sol_g[:, :n], sol_f[:, :n]) < 5e-3


@pytest.mark.parametrize("dtype,d", [(torch.float64, 3), (torch.float32, 36)])
def test_posterior_generic_path(dtype, d, dev):
    """Predictive mean and variance (exact covariance and LOVE) against the dense float64 posterior."""
    kind, n, ns, ls, os_, s2 = "rbf", 900, 150, 0.9 if d > 10 else 0.3, 1.2, 0.05
    X, y = make_data(n, d)
    Xs = torch.rand(ns, d, generator=torch.Generator().manual_seed(9), dtype=torch.float64)
    mu_ref, var_ref = OG.dense_posterior(kind, X, y, Xs, ls, os_, s2, mean=0.0)
    g, m, lik = _model(kind, X, y, ls, os_, s2, dev, dtype)
    m.eval()
    lik.eval()
    with torch.no_grad(), g.settings.max_cholesky_size(0), g.settings.eval_cg_tolerance(1e-4):
        pred = lik(m(Xs.to(dev, dtype)))
        mu, var = pred.mean, pred.variance
    assert mu.dtype == dtype
    assert rel_err(mu, mu_ref) < 1e-3
    # solve path: 150 right-hand sides stopped by the MEAN residual at eval_cg_tolerance 1e-4 -> single columns ~2e-4
    assert float((var.double().cpu() - var_ref).abs().max()) < 2e-3 * float(var_ref.abs().max())
    g2, m2, lik2 = _model(kind, X, y, ls, os_, s2, dev, dtype)
    m2.eval()
    lik2.eval()
    with torch.no_grad(), g2.settings.max_cholesky_size(0), g2.settings.fast_pred_var(True), g2.settings.max_root_decomposition_size(400 if d <= 10 else n), \
            g2.settings.eval_cg_tolerance(1e-4):
        var2 = lik2(m2(Xs.to(dev, dtype))).variance
    # LOVE is a rank-limited approximation: in d = 18 the spectrum of K decays slowly, so the full Krylov space is used
    assert float((var2.double().cpu() - var_ref).abs().max()) < 5e-3 * float(var_ref.abs().max())


@pytest.mark.parametrize("kind,d,n,m,t", [("rbf", 3, 700, 433, 1), ("rbf", 3, 1029, 1500, 11), ("matern52", 7, 515, 900, 64),
                                          ("matern32", 2, 300, 777, 65), ("matern12", 5, 260, 130, 100), ("rbf", 8, 130, 64, 17),
                                          # round 4: d <= 16 (DP = 12 / 16 instantiations, two row tiles per wave)
                                          ("rbf", 9, 700, 433, 1), ("matern52", 10, 515, 900, 65), ("rbf", 12, 300, 777, 17),
                                          ("matern32", 13, 260, 130, 100), ("matern12", 16, 1029, 1500, 11), ("rbf", 16, 130, 64, 64)])
def test_fused_float64_kernel(kind, d, n, m, t, dev):
    """kv_f64.hpp (float64 generation on the VALU + v_mfma_f64_16x16x4_f64 contraction, d <= 16) against the float64 oracle
    (1e-12) and against the row-block x DGEMM path it replaces; ragged shapes, every column-tile variant (16 / 64 / 80)."""
    from gpytorch_amd import backend as B

    g0 = torch.Generator().manual_seed(n + m + t)
    X1 = torch.rand(n, d, generator=g0, dtype=torch.float64)
    X2 = torch.rand(m, d, generator=g0, dtype=torch.float64)
    V = torch.randn(m, t, generator=g0, dtype=torch.float64)
    ls = 0.3 + 0.4 * torch.rand(d, generator=g0, dtype=torch.float64)
    K = OK.kernel_matrix(kind, X1, X2, ls.reshape(1, d), 1.3, x1_eq_x2=False, direct=True)
    shift = X1.mean(0).to(dev)
    p1 = B.prep_points(kind, X1.to(dev), ls.to(dev), shift)
    p2 = B.prep_points(kind, X2.to(dev), ls.to(dev), shift)
    assert B.fused_f64(p1, p2)
    sc = torch.tensor([1.3], dtype=torch.float64, device=dev)
    vt = B.to_probe_major(V.to(dev), torch.float64)
    out = B.from_probe_major(B.kv(p1, p2, vt, scale=sc), n)
    assert rel_err(out, K @ V) < 1e-12
    try:
        B.FORCE_CHUNKED = True
        ref = B.from_probe_major(B.kv(p1, p2, vt, scale=sc), n)
    finally:
        B.FORCE_CHUNKED = False
    assert rel_err(out, ref) < 1e-12


@pytest.mark.parametrize("dtype,d", [(torch.float64, 3), (torch.float64, 10), (torch.float32, 40)])
def test_rq_on_the_generic_path_entries_products_and_gradients(dtype, d, dev):
    """The rational-quadratic family (rq_kernel.py:61-74; oracle pinned to the reference's own forward by tests/test_oracle_golden.py)
    OUTSIDE the fused float32 kernels: float64 (fused float64 product for d <= 8, row blocks x GEMM above) and float32 with d > 32.
    Dense entries / rows / diagonal / K @ V against the float64 oracle, and the bilinear derivative with respect to the ARD
    lengthscales, the outputscale and alpha against float64 autograd."""
    from gpytorch_amd import backend as B
    from gpytorch_amd.functions import hyper_grads

    n, m, t, alpha = 420, 333, 9, 1.7
    g0 = torch.Generator().manual_seed(6)
    X1 = torch.rand(n, d, generator=g0, dtype=torch.float64)
    X2 = torch.rand(m, d, generator=g0, dtype=torch.float64)
    V = torch.randn(m, t, generator=g0, dtype=torch.float64)
    Lm = torch.randn(n, t, generator=g0, dtype=torch.float64)
    ls = ((0.9 if d > 8 else 0.3) + 0.4 * torch.rand(1, d, generator=g0, dtype=torch.float64)).requires_grad_(True)
    os_ = torch.tensor(1.7, dtype=torch.float64, requires_grad=True)
    a64 = torch.tensor(alpha, dtype=torch.float64, requires_grad=True)
    K = os_ * OK.rq(X1, X2, ls, a64, x1_eq_x2=False, direct=True)
    gl, go, ga = torch.autograd.grad((Lm * (K @ V)).sum(), [ls, os_, a64])
    K = K.detach()
    tol = 1e-11 if dtype == torch.float64 else 2e-5
    shift = X1.mean(0).to(dev, dtype)
    lsd = ls.detach().to(dev, dtype)
    p1 = B.prep_points("rq", X1.to(dev, dtype), lsd, shift, alpha)
    p2 = B.prep_points("rq", X2.to(dev, dtype), lsd, shift, alpha)
    assert p1.dtype == dtype and not p1.fused and p1.param == alpha
    sc = torch.tensor([1.7], dtype=dtype, device=dev)
    assert rel_err(B.kernel_dense(p1, p2, sc), K) < tol
    rows = torch.tensor([0, 5, n - 1, 17])
    assert rel_err(B.kernel_rows(p1, rows, p2, sc), K[rows]) < tol
    assert rel_err(B.kernel_diag(p1, p1, sc), torch.full((n,), 1.7, dtype=torch.float64)) < tol
    out = B.from_probe_major(B.kv(p1, p2, B.to_probe_major(V.to(dev), dtype), scale=sc), n)
    assert out.dtype == dtype and rel_err(out, K @ V) < tol
    kp = torch.tensor([alpha], dtype=dtype, device=dev)
    d_ls, d_os, d_al = hyper_grads(p1, p2, lsd, sc, B.to_probe_major(Lm.to(dev), dtype), B.to_probe_major(V.to(dev), dtype), kparam=kp)
    gtol = 1e-9 if dtype == torch.float64 else 1e-3
    assert rel_err(d_ls, gl) < gtol
    assert abs(float(d_os) - float(go)) < gtol * abs(float(go))
    assert abs(float(d_al) - float(ga)) < gtol * abs(float(ga)), (float(d_al), float(ga))


@pytest.mark.parametrize("dtype,d", [(torch.float64, 2), (torch.float32, 36)])
def test_rq_gp_mll_on_the_generic_path(dtype, d, dev):
    """ScaleKernel(RQKernel) ExactGP in float64 / with 18 input dimensions: the MLL and its gradients (lengthscale, alpha, outputscale,
    noise) on the Cholesky branch and on the BBMM branch with a COMPLETE probe basis (exact trace terms), against dense float64 autograd."""
    import gpytorch_amd as g

    n = 260
    X, y = make_data(n, d)
    ls0 = 0.3 if d <= 8 else 0.9

    class M(g.models.ExactGP):
        def __init__(self, x, yy, lik):
            super().__init__(x, yy, lik)
            self.mean_module = g.means.ZeroMean()
            self.covar_module = g.kernels.ScaleKernel(g.kernels.RQKernel())

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    p = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in (ls0, 2.2, 1.3, 0.1)]   # ls, alpha, outputscale, noise
    Kh = p[2] * OK.rq(X, X, p[0], p[1], x1_eq_x2=True, direct=True) + p[3] * torch.eye(n, dtype=torch.float64)
    ref = OG.dense_log_prob(Kh, y) / n
    gref = torch.autograd.grad(ref, p)
    want = torch.tensor([float(gr) * c for gr, c in zip(gref, _chain(ls0, 2.2, 1.3, 0.1 - 1e-4))], dtype=torch.float64)
    for bbmm in (False, True):
        lik = g.likelihoods.GaussianLikelihood()
        m = M(X.to(dev, dtype), y.to(dev, dtype), lik).to(dev).to(dtype)
        m.covar_module.base_kernel.lengthscale = ls0
        m.covar_module.base_kernel.alpha = 2.2
        m.covar_module.outputscale = 1.3
        lik.noise = 0.1
        mll = g.ExactMarginalLogLikelihood(lik, m)
        m.train()
        lik.train()
        if bbmm:
            Z = math.sqrt(n) * torch.eye(n, dtype=torch.float64)
            with g.settings.max_cholesky_size(0), g.settings.deterministic_probes(True), g.settings.cg_tolerance(1e-5), \
                    g.settings.max_preconditioner_size(0), g.settings.max_lanczos_quadrature_iterations(n):
                g.settings.deterministic_probes.probe_vectors = Z.to(dev)
                try:
                    val = mll(m(m.train_inputs[0]), m.train_targets)
                    val.backward()
                finally:
                    g.settings.deterministic_probes.probe_vectors = None
        else:
            val = mll(m(m.train_inputs[0]), m.train_targets)
            val.backward()
        assert val.dtype == dtype
        tol = 2e-3 if bbmm else 1e-3
        assert abs(float(val) - float(ref)) < tol * max(1.0, abs(float(ref))), (bbmm, float(val), float(ref))
        k = m.covar_module.base_kernel
        got = torch.tensor([float(k.raw_lengthscale.grad.sum()), float(k.raw_alpha.grad.sum()), float(m.covar_module.raw_outputscale.grad),
                            float(lik.noise_covar.raw_noise.grad.sum())], dtype=torch.float64)
        assert float((got - want).norm() / want.norm()) < 5 * tol, (bbmm, got, want)
