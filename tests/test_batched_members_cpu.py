"""CPU: the host logic of the stacked small-member path (``gpytorch_amd/batched.py``) -- stacking of member tensors, batched prepared
points, the chain rule from the 2 + dp derivative sums to every hyper-parameter, autograd through the stack back to batch-shaped
parameters -- with the two device launches (``gpamd_kernel_dense_batched_f32`` / ``gpamd_kernel_grad_batched_f32``) replaced by
float64 torch restatements of what ``csrc/extra_batch.hip`` computes from the PREPARED points (formulas of ``csrc/common.hpp``).
Ground truth: dense float64 autograd per member on the reference's formulas (oracle/kernels.py).  The launches themselves are
checked on the device by tests/test_gpu_batch.py::test_small_members_are_evaluated_stacked."""
import math

import pytest
import torch

from gpytorch_amd import backend as B
from gpytorch_amd import batched
from gpytorch_amd.functions import KernelSpec
from gpytorch_amd.operators import FusedKernelAddedDiagLinearOperator, FusedKernelLinearOperator
from oracle import exact_gp as OG
from oracle import kernels as OK

LN2 = math.log(2.0)


def _family(kind, s, p):
    """(k, dk/ds, dk/dp) at squared PREPARED distance s (common.hpp: cov_from_sq / dcov_dsq)."""
    if kind == "rbf":
        k = torch.exp2(-s)
        return k, -LN2 * k, torch.zeros_like(k)
    if kind == "rq":
        k = (1 + s).pow(-p)
        return k, -p * (1 + s).pow(-p - 1), -k * torch.log1p(s)
    r = s.clamp_min(0).sqrt()
    e = torch.exp(-r)
    if kind == "matern12":
        return e, torch.where(r > 1e-15, -0.5 * e / r.clamp_min(1e-300), torch.zeros_like(e)), torch.zeros_like(e)
    if kind == "matern32":
        return (1 + r) * e, -0.5 * e, torch.zeros_like(e)
    return (1 + r + s / 3) * e, -(1 + r) * e / 6, torch.zeros_like(e)


def _fake_dense(kind, zp, scale, param):
    z = zp.double()
    s = (z.unsqueeze(-2) - z.unsqueeze(-3)).pow(2).sum(-1)
    p = None if param is None else param.detach().double().reshape(-1, 1, 1)
    k = _family(kind, s, p)[0]
    return (k if scale is None else k * scale.detach().double().reshape(-1, 1, 1)).float()


def _fake_grad(kind, zp, w, param):
    z = zp.double()
    diff2 = (z.unsqueeze(-2) - z.unsqueeze(-3)).pow(2)          # [b, n, n, dp]
    s = diff2.sum(-1)
    p = None if param is None else param.detach().double().reshape(-1, 1, 1)
    k, dk, dp_ = _family(kind, s, p)
    w = w.double()
    return torch.cat([(w * k).sum((-1, -2)).unsqueeze(-1), ((w * dk).unsqueeze(-1) * diff2).sum((-2, -3)), (w * dp_).sum((-1, -2)).unsqueeze(-1)], -1)


@pytest.mark.parametrize("kind,ard", [("rbf", False), ("matern52", True), ("matern32", False), ("matern12", False), ("rq", True)])
def test_stacked_cholesky_branch_host_logic(kind, ard, monkeypatch):
    monkeypatch.setattr(batched, "kernel_dense_batched", _fake_dense)
    monkeypatch.setattr(batched, "kernel_grad_batched", _fake_grad)
    monkeypatch.setattr(B, "_require_gpu", lambda t, name: None)
    b, n, d, c = 4, 60, 3, 2
    gen = torch.Generator().manual_seed(0)
    X = torch.rand(b, n, d, generator=gen)
    Y = torch.randn(b, n, c, generator=gen)
    nls = d if ard else 1
    # batch-shaped leaves, sliced per member as kernels.py does
    ls = (0.3 + 0.5 * torch.rand(b, 1, nls, generator=gen)).requires_grad_(True)
    os_ = (0.7 + torch.rand(b, generator=gen)).requires_grad_(True)
    nz = (0.05 + 0.2 * torch.rand(b, 1, generator=gen)).requires_grad_(True)
    al = (0.8 + 2.0 * torch.rand(b, 1, generator=gen)).requires_grad_(True)
    ops = []
    for i in range(b):
        shift = X[i].mean(0) if kind.startswith("matern") else None
        spec = KernelSpec(kind, shift, param=al[i] if kind == "rq" else None)
        ops.append(FusedKernelAddedDiagLinearOperator(FusedKernelLinearOperator(X[i], X[i], spec, ls[i], os_[i].reshape(1)), nz[i]))
    iq, ld = batched.batched_inv_quad_logdet(ops, [Y[i] for i in range(b)])
    assert iq.shape == (b, c) and ld.shape == (b,)
    gi = torch.randn(b, c, generator=gen)
    gl = torch.randn(b, generator=gen)
    leaves = [ls, os_, nz] + ([al] if kind == "rq" else [])
    got = torch.autograd.grad((iq * gi).sum() + (ld * gl).sum(), leaves)
    for i in range(b):
        p = [ls[i].detach().double().requires_grad_(True), os_[i].detach().double().requires_grad_(True), nz[i].detach().double().reshape(()).requires_grad_(True),
             al[i].detach().double().reshape(()).requires_grad_(True)]
        Xi = X[i].double()
        Kd = OK.rq(Xi, Xi, p[0], p[3], x1_eq_x2=True, direct=True) if kind == "rq" else OK.kernel_matrix(kind, Xi, Xi, p[0], 1.0, x1_eq_x2=True, direct=True)
        Kh = p[1] * Kd + p[2] * torch.eye(n, dtype=torch.float64)
        Yi = Y[i].double()
        riq = (Yi * torch.linalg.solve(Kh, Yi)).sum(0)
        rld = torch.logdet(Kh)
        assert torch.allclose(iq[i].double(), riq.detach(), rtol=2e-4), (kind, i)
        assert abs(float(ld[i].detach()) - float(rld)) < 2e-4 * abs(float(rld)) + 1e-3
        want = torch.autograd.grad((riq * gi[i].double()).sum() + rld * gl[i].double(), p[:4] if kind == "rq" else p[:3])
        for q, w_ in enumerate(want):
            g_ = got[q][i].double().reshape(-1)
            w_ = w_.reshape(-1)
            assert torch.allclose(g_, w_, rtol=2e-3, atol=2e-3 * float(w_.abs().max()) + 1e-6), (kind, i, q, g_, w_)


def test_stack_prepared_is_the_prep_points_arithmetic():
    """z = coef (x - shift) / lengthscale, zero padded to dp (misc_kernels.hpp::prep_points_kernel; RQ: coef = 1 / sqrt(2 alpha))."""
    gen = torch.Generator().manual_seed(1)
    b, n, d = 3, 7, 5
    x = torch.rand(b, n, d, generator=gen)
    ls = 0.2 + torch.rand(b, d, generator=gen)
    sh = torch.rand(b, d, generator=gen)
    al = 0.5 + torch.rand(b, generator=gen)
    z = batched.stack_prepared("matern52", x, ls, sh, None)
    assert z.shape == (b, n, 8) and float(z[..., d:].abs().max()) == 0.0
    assert torch.allclose(z[..., :d], (x - sh.unsqueeze(1)) * (math.sqrt(5.0) / ls).unsqueeze(1), rtol=1e-6)
    z = batched.stack_prepared("rq", x, ls[:, :1], None, al)
    assert torch.allclose(z[..., :d], x / (ls[:, :1] * torch.sqrt(2 * al).unsqueeze(-1)).unsqueeze(1), rtol=1e-6)
    z = batched.stack_prepared("rbf", x, ls, None, None)
    assert torch.allclose(z[..., :d], x * (B.RBF_PREP_COEF / ls).unsqueeze(1), rtol=1e-6)


def test_members_stackable_policy():
    """Only single-kernel + homoskedastic-noise members on the device qualify; anything else keeps the launch plan over members."""
    from gpytorch_amd import settings

    x = torch.rand(10, 2)
    ls, nz = torch.tensor([[0.5]]), torch.tensor([0.1])
    mk = lambda xx=x, kind="rbf": FusedKernelAddedDiagLinearOperator(FusedKernelLinearOperator(xx, xx, KernelSpec(kind), ls), nz)   # noqa: E731
    assert not batched.members_stackable([mk(), mk()])            # host tensors: the path is device-only
    assert not batched.members_stackable([mk()])                  # a single member gains nothing
    with settings.batched_small_members(False):
        assert not batched.members_stackable([mk(), mk()])


def test_stacked_members_with_fixed_heteroskedastic_noise(monkeypatch):
    """FixedNoiseGaussianLikelihood members (a fixed per-point noise vector on top of the learnable scalar): the vector rides on the
    diagonal of every member's K_hat, the scalar keeps its gradient (= trace of W)."""
    monkeypatch.setattr(batched, "kernel_dense_batched", _fake_dense)
    monkeypatch.setattr(batched, "kernel_grad_batched", _fake_grad)
    monkeypatch.setattr(B, "_require_gpu", lambda t, name: None)
    b, n, d = 3, 40, 2
    gen = torch.Generator().manual_seed(5)
    X = torch.rand(b, n, d, generator=gen)
    Y = torch.randn(b, n, 1, generator=gen)
    ls = (0.4 + 0.3 * torch.rand(b, 1, 1, generator=gen)).requires_grad_(True)
    nz = (0.05 + 0.1 * torch.rand(b, 1, generator=gen)).requires_grad_(True)
    fixed = 0.02 + 0.2 * torch.rand(b, n, generator=gen)
    ops = [FusedKernelAddedDiagLinearOperator(FusedKernelLinearOperator(X[i], X[i], KernelSpec("rbf"), ls[i]), nz[i], noise_vec=fixed[i]) for i in range(b)]
    iq, ld = batched.batched_inv_quad_logdet(ops, [Y[i] for i in range(b)])
    got = torch.autograd.grad(iq.sum() + ld.sum(), [ls, nz])
    for i in range(b):
        p = [ls[i].detach().double().requires_grad_(True), nz[i].detach().double().reshape(()).requires_grad_(True)]
        Xi, Yi = X[i].double(), Y[i].double()
        Kh = OK.kernel_matrix("rbf", Xi, Xi, p[0], 1.0, x1_eq_x2=True, direct=True) + torch.diag(p[1] + fixed[i].double())
        ref = (Yi * torch.linalg.solve(Kh, Yi)).sum() + torch.logdet(Kh)
        want = torch.autograd.grad(ref, p)
        assert abs(float(iq[i].sum().detach() + ld[i].detach()) - float(ref)) < 2e-4 * abs(float(ref)) + 1e-3
        for q in range(2):
            assert abs(float(got[q][i].sum()) - float(want[q].sum())) < 3e-3 * abs(float(want[q].sum())) + 1e-5, (i, q)


@pytest.mark.parametrize("reduce_inv_quad", [True, False])
def test_batch_operator_routes_a_two_dimensional_batch_through_the_stack(reduce_inv_quad, monkeypatch):
    """``BatchLinearOperator.inv_quad_logdet`` with batch_shape [3, 2] (the reference's multi-batch mode, gpytorch/test/model_test_case.py):
    members are stacked row-major, results come back in the batch shape; log-det alone works through a dummy column."""
    from gpytorch_amd.operators import BatchLinearOperator

    monkeypatch.setattr(batched, "kernel_dense_batched", _fake_dense)
    monkeypatch.setattr(batched, "kernel_grad_batched", _fake_grad)
    monkeypatch.setattr(batched, "members_stackable", lambda ops: True)
    monkeypatch.setattr(B, "_require_gpu", lambda t, name: None)
    bs, n, d, c = (3, 2), 30, 2, 2
    gen = torch.Generator().manual_seed(2)
    X = torch.rand(*bs, n, d, generator=gen)
    Y = torch.randn(*bs, n, c, generator=gen)
    ls = 0.3 + 0.5 * torch.rand(*bs, 1, 1, generator=gen)
    nz = 0.05 + 0.2 * torch.rand(*bs, 1, generator=gen)
    ops = [FusedKernelAddedDiagLinearOperator(FusedKernelLinearOperator(X[i, j], X[i, j], KernelSpec("rbf"), ls[i, j]), nz[i, j]) for i in range(3) for j in range(2)]
    op = BatchLinearOperator(ops, torch.Size(bs))
    iq, ld = op.inv_quad_logdet(Y, logdet=True, reduce_inv_quad=reduce_inv_quad)
    assert ld.shape == bs and iq.shape == (bs if reduce_inv_quad else (*bs, c))
    for i in range(3):
        for j in range(2):
            Kh = OK.kernel_matrix("rbf", X[i, j].double(), X[i, j].double(), ls[i, j].double(), 1.0, x1_eq_x2=True, direct=True) + float(nz[i, j]) * torch.eye(n, dtype=torch.float64)
            riq = (Y[i, j].double() * torch.linalg.solve(Kh, Y[i, j].double())).sum(0)
            want = riq.sum() if reduce_inv_quad else riq
            assert torch.allclose(iq[i, j].double(), want, rtol=2e-4)
            assert abs(float(ld[i, j]) - float(torch.logdet(Kh))) < 2e-4 * abs(float(torch.logdet(Kh))) + 1e-3
    iq0, ld0 = op.inv_quad_logdet(None, logdet=True)
    assert iq0 is None and torch.allclose(ld0, ld)
