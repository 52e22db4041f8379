"""Fused float32 kernels for 17 .. 32 input dimensions (round 5; D = 20 / 24 / 32 instantiations, ``csrc/kv_dispatch.hpp`` ``kv_kernel_dims``).

The reference has no dimension limit (``gpytorch/kernels/kernel.py:26-49``: the Gram-trick ``sq_dist`` is a GEMM whatever d; the KeOps
precedent ``gpytorch/kernels/keops/rbf_kernel.py:44-55`` reduces over any d as well); until round 4 inputs beyond 16 dimensions dropped to
dense row blocks x library GEMM (13-24x slower).  Covered here against the float64 oracle, through the C ABI: every column-count
kernel of the three generation paths (direct differences, Gram form with the fp32 contraction, Gram form with the split contraction) at
d = 17, 20, 24, 26 (stride 32: ``backend.padded_dim``) and 32; explicit rows / diagonal; the bilinear derivative (single lengthscale on
the Gram-form kernel -- fp32 and split W --, per-dimension sums on the direct kernel, input gradients on row blocks); the MLL with gradients
and the posterior through the model API; and one full-chip shape per path (n = 40 000: every SIMD busy, the occupancy-1 instantiations)."""
import math

import pytest
import torch

from oracle import exact_gp as OG
from oracle import kernels as OK
from tests.util import make_data, rel_err

pytestmark = pytest.mark.gpu


def _dense(kind, x1, x2, ls, alpha=None):
    """float64 covariance block from pairwise distances (the KeOps formulas, keops/rbf_kernel.py:12-15, keops/matern_kernel.py:13-30; RQ:
    rq_kernel.py:61-74) on whatever device the inputs live."""
    r2 = torch.cdist(x1 / ls, x2 / ls).pow(2)
    if kind == "rbf":
        return torch.exp(-0.5 * r2)
    if kind == "rq":
        return (1.0 + r2 / (2.0 * alpha)).pow(-alpha)
    nu = OK.KINDS[kind]
    r = r2.sqrt() * math.sqrt(2 * nu)
    e = torch.exp(-r)
    return e if nu == 0.5 else ((1 + r) * e if nu == 1.5 else (1 + r + r * r / 3) * e)


def _clouds(n, m, d, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n, d, generator=g, dtype=torch.float64), torch.rand(m, d, generator=g, dtype=torch.float64), g


@pytest.mark.parametrize("kind,d", [("rbf", 17), ("rbf", 20), ("matern52", 24), ("rq", 24), ("matern32", 26), ("rbf", 32), ("matern52", 32), ("matern12", 20)])
def test_kv_every_column_count_on_every_path(kind, d, dev):
    from gpytorch_amd import backend as B

    n, m = 700, 900
    X1, X2, g = _clouds(n, m, d, 100 + d)
    ls = 0.2 + 0.08 * d
    alpha = 1.3 if kind == "rq" else None
    shift = X1.mean(0).float().to(dev)
    p1 = B.prep_points(kind, X1.float().to(dev), torch.tensor(ls), shift, param=alpha)
    p2 = B.prep_points(kind, X2.float().to(dev), torch.tensor(ls), shift, param=alpha)
    assert p1.fused and p1.dp == (32 if d > 24 else 4 * ((d + 3) // 4))
    K = _dense(kind, X1, X2, ls, alpha)
    assert (B.kernel_dense(p1, p2).double().cpu() - K).abs().max() < 4e-6
    rows = torch.tensor([0, n - 1, 123])
    assert (B.kernel_rows(p1, rows, p2).double().cpu() - K[rows]).abs().max() < 4e-6
    assert torch.equal(B.kernel_diag(p1, p1).cpu(), torch.ones(n))
    for t in (1, 2, 4, 7, 11, 16, 17, 33, 64, 65, 70, 129):
        V = torch.randn(m, t, generator=g, dtype=torch.float64)
        vt = B.to_probe_major(V.to(dev))
        ref = K @ V
        paths = [(0, None)] if kind in ("matern12",) else [(0, None), (B.KV_GRAM, False), (B.KV_GRAM, True)]
        if kind == "rq":
            paths = paths[1:]            # (the rational-quadratic family has no direct-difference product kernel: Gram form or row blocks)
        try:
            for flags, split in paths:
                B.FORCE_KV_FLAGS = flags | (B.KV_SPLIT if split else 0)
                out = B.from_probe_major(B.kv(p1, p2, vt), n)
                assert rel_err(out, ref) < (2e-5 if flags == 0 else 5e-5), (kind, d, t, flags, split)
        finally:
            B.FORCE_KV_FLAGS = None
    # the automatic policy picks the Gram form here (cloud-centred: max |z|^2 <= 32)
    if kind != "matern12":
        assert B.kv_flags(p1, p2, 65) & B.KV_GRAM


@pytest.mark.parametrize("kind,d,t", [("rbf", 24, 65), ("matern52", 32, 64), ("rbf", 32, 11), ("matern32", 20, 1), ("rbf", 32, 2)])
def test_kv_on_a_full_chip(kind, d, t, dev):
    """n = 40 000 rows: every CU holds its full complement of workgroups (the D = 24 / 32 split kernels at ONE wave per SIMD, the 93 KB LDS image
    of D = 32), both contraction paths and the direct kernels against float64 rows computed on the device."""
    from gpytorch_amd import backend as B

    n = 40_000
    g = torch.Generator().manual_seed(7 * d + t)
    X = torch.rand(n, d, generator=g)
    ls = 0.2 + 0.08 * d
    xp = B.prep_points(kind, X.to(dev), torch.tensor(ls), X.mean(0).to(dev))
    V = torch.randn(t, B.round_up(n, 4), generator=g).to(dev)
    rows = torch.cat([torch.arange(0, 300), torch.arange(n - 300, n), torch.randint(0, n, (400,), generator=g)])
    Xd = X.double().to(dev)
    Kr = _dense(kind, Xd[rows.to(dev)], Xd, ls)
    ref = (Kr @ V[:, :n].double().t()).cpu()
    try:
        for flags in (0, B.KV_GRAM, B.KV_GRAM | B.KV_SPLIT):
            B.FORCE_KV_FLAGS = flags
            for _ in range(2):
                out = B.kv(xp, xp, V)[:, :n].t()[rows.to(dev)]
                assert rel_err(out, ref) < (2e-5 if flags == 0 else 5e-5), (kind, d, t, flags)
    finally:
        B.FORCE_KV_FLAGS = None


@pytest.mark.parametrize("kind,d,t", [("rbf", 24, 11), ("matern52", 20, 30), ("rbf", 32, 65), ("matern32", 26, 40)])
def test_bilinear_derivative_beyond_16_dimensions(kind, d, t, dev):
    """Single lengthscale: the Gram-form derivative kernel (kv_grad2 MODE 0; W on the fp32 MFMAs below 24 columns, split above); ARD: the
    direct-difference kernel's per-dimension sums; input gradients: row blocks.  Against float64 autograd of the dense formulas."""
    from gpytorch_amd import backend as B
    from gpytorch_amd.functions import hyper_grads

    n, m_ = 600, 777
    X1, X2, g0 = _clouds(n, m_, d, 50 + d)
    Lm = torch.randn(n, t, generator=g0, dtype=torch.float64).abs()
    Rm = torch.randn(m_, t, generator=g0, dtype=torch.float64).abs()
    os_ = torch.tensor(1.7, dtype=torch.float64, requires_grad=True)
    for ard in (False, True):
        ls = ((0.2 + 0.08 * d) * (1.0 + 0.3 * torch.rand(1, d if ard else 1, generator=g0, dtype=torch.float64))).requires_grad_(True)
        x1 = X1.clone().requires_grad_(True)
        K = OK.kernel_matrix(kind, x1, X2, ls.expand(1, d), os_, x1_eq_x2=False, direct=True)
        gl, go, gx = torch.autograd.grad((Lm * (K @ Rm)).sum(), [ls, os_, x1])
        shift = X1.mean(0).float().to(dev)
        lsd = ls.detach().float().to(dev)
        p1 = B.prep_points(kind, X1.float().to(dev), lsd, shift)
        p2 = B.prep_points(kind, X2.float().to(dev), lsd, shift)
        assert p1.fused and B.grad_gram_ok(p1, p2)
        lt, rt = B.to_probe_major(Lm.to(dev)), B.to_probe_major(Rm.to(dev))
        d_ls, d_os = hyper_grads(p1, p2, lsd, os_.detach().float().to(dev).reshape(1), lt, rt)
        assert rel_err(d_ls, gl) < 1e-3, (kind, d, ard)
        assert abs(float(d_os) - float(go)) < 1e-3 * abs(float(go))
        if not ard:
            out = hyper_grads(p1, p2, lsd, os_.detach().float().to(dev).reshape(1), lt, rt, want_x1=True)
            assert rel_err(out[0], gl) < 1e-3 and rel_err(out[2], gx) < 1e-3, (kind, d)


@pytest.mark.parametrize("kind,d", [("matern52", 24), ("rbf", 32)])
def test_mll_and_grads_beyond_16_dimensions(kind, d, dev):
    """The model API on the fused path at d = 24 / 32: Cholesky branch and BBMM branch (complete probe basis: deterministic) against the dense
    float64 MLL and its gradients -- the check ``tests/test_gpu_generic.py`` runs for the row-block path these dimensions took before."""
    from tests.test_gpu_generic import _chain, _model

    n, ls, os_, s2 = 260, 0.9, 1.4, 0.1
    X, y = make_data(n, d)
    ref, gref = OG.dense_mll_and_grads(kind, X, y, ls, os_, s2, mean=0.1)
    c = _chain(ls, os_, s2 - 1e-4)
    for bbmm in (False, True):
        g, m, lik = _model(kind, X, y, ls, os_, s2, dev, torch.float32, mean=0.1)
        mll = g.ExactMarginalLogLikelihood(lik, m)
        m.train(), lik.train()
        if bbmm:
            Z = math.sqrt(n) * torch.eye(n, dtype=torch.float64)
            with g.settings.max_cholesky_size(0), g.settings.deterministic_probes(True), g.settings.cg_tolerance(1e-5), \
                    g.settings.max_preconditioner_size(0), g.settings.max_lanczos_quadrature_iterations(n):
                g.settings.deterministic_probes.probe_vectors = Z.to(dev)
                try:
                    op = lik(m(m.train_inputs[0])).lazy_covariance_matrix.evaluate_kernel()
                    assert op.kernel_op.prepared()[0].fused
                    val = mll(m(m.train_inputs[0]), m.train_targets)
                    val.backward()
                finally:
                    g.settings.deterministic_probes.probe_vectors = None
        else:
            val = mll(m(m.train_inputs[0]), m.train_targets)
            val.backward()
        tol = 2e-3 if bbmm else 1e-3
        assert abs(float(val) - float(ref)) < tol * max(1.0, abs(float(ref))), (bbmm, float(val), float(ref))
        got = (m.covar_module.base_kernel.raw_lengthscale.grad, m.covar_module.raw_outputscale.grad, lik.noise_covar.raw_noise.grad)
        for gg, rr, cc in zip(got, gref, c):
            assert abs(float(gg.sum()) - float(rr) * cc) < 5 * tol * abs(float(rr) * cc) + 2e-5, (kind, bbmm, float(gg.sum()), float(rr) * cc)


def test_posterior_at_20_dimensions(dev):
    from tests.test_gpu_generic import _model

    kind, d, n, ns, ls, os_, s2 = "rbf", 20, 900, 150, 0.9, 1.2, 0.05
    X, y = make_data(n, d)
    Xs = torch.rand(ns, d, generator=torch.Generator().manual_seed(9), dtype=torch.float64)
    mu_ref, var_ref = OG.dense_posterior(kind, X, y, Xs, ls, os_, s2, mean=0.0)
    g, m, lik = _model(kind, X, y, ls, os_, s2, dev, torch.float32)
    m.eval(), lik.eval()
    with torch.no_grad(), g.settings.max_cholesky_size(0), g.settings.eval_cg_tolerance(1e-4):
        pred = lik(m(Xs.float().to(dev)))
    assert rel_err(pred.mean, mu_ref) < 1e-3
    assert float((pred.variance.double().cpu() - var_ref).abs().max()) < 2e-3 * float(var_ref.abs().max())
