"""Batches of SMALL independent exact GPs: the Cholesky branch of the marginal log likelihood for all members at once.

Reference behaviour: a batch-mode ``ExactGP`` (``gpytorch/kernels/kernel.py:163-208`` ``batch_shape``,
``test/examples/test_batch_gp_regression.py``) evaluates ``K(X, X)`` as one ``[*batch, n, n]`` tensor and -- below
``settings.max_cholesky_size`` (``gpytorch/settings.py``: 800) -- factorises it with one batched Cholesky
(``distributions/multivariate_normal.py:249`` through ``inv_quad_logdet``).  The fused kernels of this library work on one point cloud at
a time, so :class:`~gpytorch_amd.operators.BatchLinearOperator` is a launch plan over members; for members this small nothing but
launches is left of that plan (~50 per member and evaluation).  Here the members are stacked instead:

* forward: prepared points of all members ``[b, n, dp]`` (three elementwise torch ops), ``gpamd_kernel_dense_batched_f32`` (one launch,
  ``blockIdx.z`` = member), float64 batched Cholesky + ONE batched triangular solve (L^-1) through torch (rocSOLVER / rocBLAS);
* backward: ``W_g = g_ld K_g^-1 - sum_c g_iq[c] a_gc a_gc^T`` for every member in float64 (K^-1 = L^-T L^-1 from the forward's triangular solve + one ``baddbmm``),
  then ``gpamd_kernel_grad_batched_f32`` (one launch) reduces ``sum W dK/dtheta`` of every member to the 2 + dp numbers the
  hyper-parameter chain rule needs (the same convention as ``backend.kv_grad2``).

Used when every member is a single stationary kernel + homoskedastic noise (optionally + a FIXED heteroskedastic noise vector) on the fused float32 path with the same family and shape
(``members_stackable``); everything else keeps the member loop."""
from __future__ import annotations


import torch

from . import backend as B
from . import settings
from ._lib import check, lib
from .operators import psd_safe_cholesky


def stack_prepared(kind: str, x: torch.Tensor, ls: torch.Tensor, shift, param) -> torch.Tensor:
    """``gpamd_prep_points_f32`` for b clouds at once: z = coef (x - shift) / lengthscale, zero padded to dp -> [b, n, dp] float32.
    x [b, n, d]; ls [b, 1 | d]; shift [b, d] or None; param [b] (RQ alpha) or None."""
    b, n, d = x.shape
    dp = B.padded_dim(d)
    coef = (1.0 / torch.sqrt(2.0 * param.detach().to(torch.float32))).reshape(b, 1) if kind == "rq" else B.prep_coef(kind)
    mul = (coef / ls.detach().to(torch.float32).reshape(b, -1)).reshape(b, 1, -1)
    xs = x.detach().to(torch.float32)
    if shift is not None:
        xs = xs - shift.detach().to(torch.float32).reshape(b, 1, d)
    zp = torch.zeros(b, n, dp, device=x.device, dtype=torch.float32)
    zp[..., :d] = xs * mul
    return zp


def kernel_dense_batched(kind: str, zp: torch.Tensor, scale, param) -> torch.Tensor:
    """[b, n, n] float32: scale[g] * k(z_g, z_g) for every member in one launch."""
    b, n, dp = zp.shape
    out = torch.empty(b, n, n, device=zp.device, dtype=torch.float32)
    kp = None if param is None else param.detach().to(torch.float32).reshape(b).contiguous()
    sc = None if scale is None else scale.detach().to(torch.float32).reshape(b).contiguous()
    check(lib().gpamd_kernel_dense_batched_f32(B.KIND_IDS[kind], B._ptr(kp), B._ptr(zp), n, B._ptr(zp), n, dp, b, B._ptr(sc), None, B._ptr(out),
                                               n, B._stream(zp.device)), "kernel_dense_batched")
    return out


def kernel_grad_batched(kind: str, zp: torch.Tensor, w: torch.Tensor, param) -> torch.Tensor:
    """[b, 2 + dp] float64: (sum W k | sum W dk/ds (z_iq - z_jq)^2 per prepared dimension | sum W dk/dparam) of every member in one launch."""
    b, n, dp = zp.shape
    w = w.to(torch.float32).contiguous()
    g = torch.empty(b, 2 + dp, device=zp.device, dtype=torch.float64)
    kp = None if param is None else param.detach().to(torch.float32).reshape(b).contiguous()
    check(lib().gpamd_kernel_grad_batched_f32(B.KIND_IDS[kind], B._ptr(kp), B._ptr(zp), n, B._ptr(zp), n, dp, b, B._ptr(w), w.stride(1), B._ptr(g),
                                              B._stream(zp.device)), "kernel_grad_batched")
    return g


class BatchedCholeskyInvQuadLogdetFn(torch.autograd.Function):
    """(inv_quad [b, c], logdet [b]) of K_hat_g = outputscale_g k(x_g, x_g; lengthscale_g) + noise_g I for b stacked members;
    the batched twin of ``functions.CholeskyInvQuadLogdetFn`` (same arithmetic: float32 generation, float64 factorisation)."""

    @staticmethod
    def forward(ctx, x, lengthscale, outputscale, noise, rhs, kind, shift, kparam, noise_vec=None):
        B._require_gpu(x, "x")
        zp = stack_prepared(kind, x, lengthscale, shift, kparam)
        K = kernel_dense_batched(kind, zp, outputscale, kparam).to(torch.float64)
        K.diagonal(dim1=-2, dim2=-1).add_(noise.detach().to(torch.float64).reshape(-1, 1))
        if noise_vec is not None:   # fixed heteroskedastic noise (FixedNoiseGaussianLikelihood): [b, n], not learnable
            K.diagonal(dim1=-2, dim2=-1).add_(noise_vec.detach().to(torch.float64))
        Lc = psd_safe_cholesky(K, model_dtype=rhs.dtype)   # jitter only for the members whose plain factorisation fails
        r64 = rhs.detach().to(torch.float64)
        # L^-1 once (one batched triangular solve, rocBLAS; reused by the backward for K^-1): the batched potrs behind torch.cholesky_solve
        # returned hipErrorLaunchFailure for [2, 600, 600] float64 on this stack (profiles/r03_s23_*), the triangular solve has no size limit
        Linv = torch.linalg.solve_triangular(Lc, torch.eye(K.shape[-1], device=K.device, dtype=torch.float64).expand_as(K), upper=False)
        sol = Linv.transpose(-1, -2) @ (Linv @ r64)
        inv_quad = (sol * r64).sum(-2)
        logdet = 2.0 * Lc.diagonal(dim1=-2, dim2=-1).log().sum(-1)
        ctx.kind, ctx.zp, ctx.d = kind, zp, x.shape[-1]
        ctx.has_os, ctx.has_par = outputscale is not None, kparam is not None
        empty = torch.empty(0, device=x.device)
        ctx.save_for_backward(lengthscale, outputscale if ctx.has_os else empty, kparam if ctx.has_par else empty, Linv, sol)
        return inv_quad.to(rhs.dtype), logdet.to(rhs.dtype)

    @staticmethod
    def backward(ctx, g_iq, g_ld):
        lengthscale, outputscale, kparam, Linv, sol = ctx.saved_tensors
        zp, d = ctx.zp, ctx.d
        b, n, dp = zp.shape
        g_iq = g_iq.to(torch.float64)
        # d logdet = tr(K^-1 dK);  d inv_quad[c] = -a_c^T dK a_c
        w = (Linv.transpose(-1, -2) @ Linv) * g_ld.to(torch.float64).reshape(b, 1, 1)
        w = torch.baddbmm(w, sol * g_iq.reshape(b, 1, -1), sol.transpose(-1, -2), alpha=-1.0)
        d_noise = w.diagonal(dim1=-2, dim2=-1).sum(-1)
        G = kernel_grad_batched(ctx.kind, zp, w, kparam if ctx.has_par else None)
        theta = outputscale.detach().to(torch.float64).reshape(b) if ctx.has_os else torch.ones(b, device=zp.device, dtype=torch.float64)
        ls = lengthscale.detach().to(torch.float64).reshape(b, -1)
        gq = G[:, 1 : 1 + d]
        if ls.shape[1] == 1:
            d_ls = theta.unsqueeze(-1) * (-2.0) / ls * gq.sum(-1, keepdim=True)
        else:
            d_ls = theta.unsqueeze(-1) * (-2.0) / ls * gq
        d_os = G[:, 0].reshape(outputscale.shape).to(outputscale.dtype) if ctx.has_os else None
        d_par = None
        if ctx.has_par and ctx.needs_input_grad[7]:
            # RQ: s = |dx|^2 / (2 alpha l^2) -> dK/dalpha = dk/dalpha|_s - dk/ds s / alpha (functions.hyper_grads)
            d_par = (theta * (G[:, 1 + dp] - gq.sum(-1) / kparam.detach().to(torch.float64).reshape(b))).reshape(kparam.shape).to(kparam.dtype)
        d_rhs = (2.0 * sol * g_iq.reshape(b, 1, -1)).to(g_ld.dtype) if ctx.needs_input_grad[4] else None
        return None, d_ls.reshape(lengthscale.shape).to(lengthscale.dtype), d_os, d_noise.reshape(-1).to(g_ld.dtype), d_rhs, None, None, d_par, None


def members_stackable(ops) -> bool:
    """Every member is ``outputscale * k(x, x) + noise I`` with ONE stationary kernel of the same family on the fused float32 path
    (d <= 16), the same number of points, lengthscales and optional parts (output scale, centring shift, shape parameter, fixed
    heteroskedastic noise vector), no gradient with respect to the inputs -- and small enough for the Cholesky branch."""
    from .operators import FusedKernelAddedDiagLinearOperator

    if len(ops) < 2 or settings.batched_small_members.off():
        return False
    o0 = ops[0]
    if type(o0) is not FusedKernelAddedDiagLinearOperator:
        return False
    if not o0._use_cholesky(settings.fast_computations.log_prob):
        # mid-size members (settings.batched_small_members.max_size): stacked dense evaluation instead of one BBMM evaluation per member,
        # as long as the float64 [b, n, n] work arrays (covariances, factors, inverse) fit the budget
        n = o0.shape[-1]
        if not (n <= settings.batched_small_members.max_size and 3.0 * 8.0 * len(ops) * n * n <= settings.batched_small_members.max_bytes
                and settings.max_cholesky_size.value() >= 800 and settings.fast_computations.log_prob.on()):
            return False   # (a max_cholesky_size LOWERED by the user -- e.g. 0 to force mBCG -- is respected)
    k0 = o0.kernel_op
    if B.work_dtype(k0.x1) != torch.float32 or k0.x1.shape[-1] > 16 or not k0.x1.is_cuda:   # (csrc/extra_batch.hip GB_MAXDP: the stacked kernels hold a point in 16 registers)
        return False
    for o in ops:
        if type(o) is not FusedKernelAddedDiagLinearOperator or (o.noise_vec is None) != (o0.noise_vec is None):
            return False
        if o.noise_vec is not None and (o.noise_vec.requires_grad or o.noise_vec.numel() != o0.shape[-1]):
            return False
        k = o.kernel_op
        if (k.spec.kind != k0.spec.kind or k.x1.shape != k0.x1.shape or k.x1.dtype != k0.x1.dtype or not k.square_same_inputs
                or k.x1.requires_grad or k.x2.requires_grad or k.lengthscale.numel() != k0.lengthscale.numel()
                or (k.outputscale is None) != (k0.outputscale is None) or (k.spec.shift is None) != (k0.spec.shift is None)
                or (k.spec.param is None) != (k0.spec.param is None) or k.spec.dvec is not None):
            return False
    return True


def _stack(ts):
    """torch.stack, or one expand when the batch repeats a single tensor (shared inputs / hyper-parameters)."""
    t0 = ts[0]
    if all(t is t0 for t in ts):
        return t0.unsqueeze(0).expand(len(ts), *t0.shape)
    return torch.stack(list(ts), 0)


def batched_inv_quad_logdet(ops, rhs_members):
    """(inv_quad [b, c], logdet [b]) of stackable members (``members_stackable``); ``rhs_members``: one [n, c] tensor per member."""
    ks = [o.kernel_op for o in ops]
    k0 = ks[0]
    b = len(ops)
    x = _stack([k.x1 for k in ks])
    ls = _stack([k.lengthscale.reshape(-1) for k in ks])
    os_ = None if k0.outputscale is None else _stack([k.outputscale.reshape(()) for k in ks])
    noise = _stack([o.noise.reshape(()) for o in ops])
    shift = None if k0.spec.shift is None else _stack([k.spec.shift.reshape(-1) for k in ks])
    par = None if k0.spec.param is None else _stack([k.spec.param.reshape(()) for k in ks])
    rhs = _stack(list(rhs_members))
    assert rhs.shape[0] == b
    nvec = None if ops[0].noise_vec is None else _stack([o.noise_vec.reshape(-1) for o in ops])
    return BatchedCholeskyInvQuadLogdetFn.apply(x, ls, os_, noise, rhs, k0.spec.kind, shift, par, nvec)
