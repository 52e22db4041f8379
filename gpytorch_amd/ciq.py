"""Contour-integral quadrature (CIQ) with multi-shift MINRES: ``K^{-1/2} b`` and ``K^{1/2} b`` through matrix-free products.

Mirrors ``gpytorch.sqrt_inv_matmul`` (``gpytorch/__init__.py:252-278``) -> ``LinearOperator.sqrt_inv_matmul`` ->
``linear_operator.utils.contour_integral_quad`` + ``linear_operator.utils.minres`` (third party, not vendored; algorithm of
Pleiss et al., "Fast matrix square roots with applications to Gaussian processes and Bayesian optimization", NeurIPS 2020, and
Hale, Higham & Trefethen, "Computing A^alpha, log(A) and related matrix functions by contour integrals", SIAM J. Numer. Anal. 2008,
method 3).  Consumers in the reference: ``variational/ciq_variational_strategy.py:217`` and posterior sampling
(``examples/02_Scalable_Exact_GPs/Exact_GP_Posterior_Sampling_with_CIQ.ipynb``).

    K^{-1/2} ~= sum_q w_q (K + s_q I)^{-1},      s_q = lmin * sc(u_q | k')^2,   w_q = 2 K' sqrt(lmin) / (pi Q) * dn(u_q | k') / cn(u_q | k')^2,
    u_q = (q - 1/2) K' / Q,   k'^2 = 1 - lmin / lmax,   K' = K(k')  (complete elliptic integral)

All Q shifted systems share ONE Krylov space: msMINRES runs one Lanczos recurrence on K (one fused K*V per iteration, all
right-hand sides as columns) and Q sets of Givens / solution updates.  Vectors are probe-major; eigenvalue bounds come from a
short Lanczos run, as in the reference.
"""
from __future__ import annotations

import math

import torch


def ciq_weights_shifts(lmin: float, lmax: float, num_quad: int = 15):
    """(weights [Q], shifts [Q]) of the quadrature rule above (float64 CPU tensors)."""
    from scipy.special import ellipj, ellipk

    kp2 = 1.0 - lmin / lmax                      # parameter m = k'^2 of the complementary modulus
    Kp = float(ellipk(kp2))
    u = (torch.arange(1, num_quad + 1, dtype=torch.float64) - 0.5) * (Kp / num_quad)
    sn, cn, dn, _ = ellipj(u.numpy(), kp2)
    sn, cn, dn = (torch.from_numpy(a) for a in (sn, cn, dn))
    shifts = lmin * (sn / cn) ** 2
    weights = (2.0 * Kp * math.sqrt(lmin) / (math.pi * num_quad)) * dn / cn**2
    return weights, shifts


def msminres(matvec, rhs_t: torch.Tensor, shifts: torch.Tensor, n: int, tol: float | None = None, max_iter: int = 400):
    """Solve (A + shifts[q] I) x = b for all shifts and all rows b of ``rhs_t`` ([t, ld] probe-major) with one Lanczos recurrence.

    ``matvec(v [t, ld]) -> [t, ld]``.  Returns (X [Q, t, ld], iterations).  MINRES recurrences per (shift, column):
    Givens rotations applied to the shifted tridiagonal column [beta_k; alpha_k + s; beta_{k+1}], search directions
    d_k = (v_k - delta_k d_{k-1} - eps_k d_{k-2}) / gamma_k, x += tau_k d_k, residual norm |phibar_k|; stops when the largest
    relative residual over shifts and columns falls below ``tol``."""
    if tol is None:
        from . import settings

        tol = settings.minres_tolerance.value()
    dev, dt = rhs_t.device, rhs_t.dtype
    Q, t = shifts.numel(), rhs_t.shape[0]
    native = rhs_t.is_cuda and dt == torch.float32 and rhs_t.stride(1) == 1 and rhs_t.stride(0) % 4 == 0 and rhs_t.shape[1] == rhs_t.stride(0)
    if native:
        from . import backend as B
        from ._lib import check, lib
    sh = shifts.to(device=dev, dtype=dt).reshape(Q, 1)
    beta1 = rhs_t[:, :n].norm(dim=-1).clamp_min(1e-30)                     # [t]
    v = rhs_t / beta1.unsqueeze(-1)
    v_prev = torch.zeros_like(v)
    beta = beta1.clone()                                                    # beta_k (beta_1 for the first step)
    X = torch.zeros(Q, *rhs_t.shape, device=dev, dtype=dt)
    d1 = torch.zeros_like(X)                                                # d_{k-1}
    d2 = torch.zeros_like(X)                                                # d_{k-2}
    c1 = torch.ones(Q, t, device=dev, dtype=dt)                             # c_{k-1}
    s1 = torch.zeros(Q, t, device=dev, dtype=dt)
    c2 = torch.ones(Q, t, device=dev, dtype=dt)                             # c_{k-2}
    s2 = torch.zeros(Q, t, device=dev, dtype=dt)
    phibar = beta1.unsqueeze(0).expand(Q, t).clone()
    it = 0
    for it in range(1, max_iter + 1):
        w = matvec(v)
        if it > 1:
            w = w - beta.unsqueeze(-1) * v_prev
        alpha = (w[:, :n] * v[:, :n]).sum(-1)                               # [t]
        w = w - alpha.unsqueeze(-1) * v
        beta_next = w[:, :n].norm(dim=-1)                                   # [t]
        bk = beta.unsqueeze(0) if it > 1 else torch.zeros(1, t, device=dev, dtype=dt)   # beta_k couples to v_{k-1}: none at k = 1
        eps = s2 * bk
        dhat = c2 * bk
        a_s = alpha.unsqueeze(0) + sh                                       # alpha_k + shift  [Q, t]
        delta = c1 * dhat + s1 * a_s
        gbar = -s1 * dhat + c1 * a_s
        gamma = torch.sqrt(gbar * gbar + beta_next.unsqueeze(0) ** 2).clamp_min(1e-30)
        c, s = gbar / gamma, beta_next.unsqueeze(0) / gamma
        tau = c * phibar
        phibar = -s * phibar
        if native:
            # one fused pass over the [Q, t, n] state (csrc/lanczos_kernels.hpp msminres_update_kernel): the new direction overwrites d2
            coef = torch.stack([delta.expand(Q, t), eps.expand(Q, t), 1.0 / gamma, tau]).contiguous()
            vc = v.contiguous()
            check(lib().gpamd_msminres_update_f32(B._ptr(vc), B._ptr(d1), B._ptr(d2), B._ptr(X), B._ptr(coef), Q, t, n, rhs_t.stride(0),
                                                   B._stream(dev)), "msminres_update")
            d2, d1 = d1, d2
        else:
            d = (v.unsqueeze(0) - delta.unsqueeze(-1) * d1 - eps.unsqueeze(-1) * d2) / gamma.unsqueeze(-1)
            X = X + tau.unsqueeze(-1) * d
            d2, d1 = d1, d
        c2, s2, c1, s1 = c1, s1, c, s
        if float((phibar.abs() / beta1.unsqueeze(0)).max()) < tol or float(beta_next.max()) < 1e-12:
            break
        v_prev, v = v, w / beta_next.clamp_min(1e-30).unsqueeze(-1)
        beta = beta_next
    return X, it


def lanczos_eig_bounds(matvec, n: int, device, dtype, iters: int = 20, generator=None):
    """(lmin, lmax) estimates from a short Lanczos run on a random vector (as ``contour_integral_quad`` does)."""
    from . import backend as B

    ld = B.round_up(n, 4)
    q = torch.zeros(1, ld, device=device, dtype=dtype)
    q[:, :n] = torch.randn(1, n, device=device, dtype=dtype, generator=generator)
    q = q / q.norm()
    q_prev = torch.zeros_like(q)
    alphas, betas = [], []
    beta = torch.zeros((), device=device, dtype=dtype)
    for _ in range(min(iters, n)):
        w = matvec(q) - beta * q_prev
        a = (w * q).sum()
        w = w - a * q
        beta = w.norm()
        alphas.append(a)
        betas.append(beta)
        if float(beta) < 1e-10:
            break
        q_prev, q = q, w / beta
    a = torch.stack(alphas).double().cpu()
    T = torch.diag(a)
    if len(alphas) > 1:
        b = torch.stack(betas[: len(alphas) - 1]).double().cpu()
        T = T + torch.diag(b, 1) + torch.diag(b, -1)
    ev = torch.linalg.eigvalsh(T)
    return float(ev[0].clamp_min(1e-12)), float(ev[-1])


def contour_integral_quad(matvec, rhs_t, n, inverse=True, num_quad=None, lmin=None, lmax=None, tol=None, max_iter=400, generator=None):
    """K^{-1/2} rhs (``inverse``) or K^{1/2} rhs = K (K^{-1/2} rhs), rows of ``rhs_t`` probe-major.  Returns (result_t, info)."""
    from . import settings

    num_quad = settings.num_contour_quadrature.value() if num_quad is None else num_quad
    if lmin is None or lmax is None:
        lo, hi = lanczos_eig_bounds(matvec, n, rhs_t.device, rhs_t.dtype, generator=generator)
        # Ritz values lie INSIDE the spectrum: widen the interval (the rule only needs [lmin, lmax] to contain it)
        lmin = lo * 0.5 if lmin is None else lmin
        lmax = hi * 1.1 if lmax is None else lmax
    weights, shifts = ciq_weights_shifts(lmin, lmax, num_quad)
    X, iters = msminres(matvec, rhs_t, shifts, n, tol=tol, max_iter=max_iter)
    res = (weights.to(device=X.device, dtype=X.dtype).reshape(-1, 1, 1) * X).sum(0)
    if not inverse:
        res = matvec(res)
    return res, dict(iterations=iters, lmin=lmin, lmax=lmax, num_quad=num_quad)


class SqrtInvMatmulFn(torch.autograd.Function):
    """K_hat^{-1/2} rhs for K_hat = outputscale k(x, x; lengthscale) + noise I with the BACKWARD pass the reference differentiates through
    (``gpytorch/__init__.py:252-278`` "backward pass"; consumer ``variational/ciq_variational_strategy.py:217``).

        K^{-1/2} b = sum_q w_q (K + s_q I)^-1 b        =>        d/dtheta [g^T K^{-1/2} b] = - sum_q w_q ((K + s_q)^-1 g)^T (dK/dtheta) ((K + s_q)^-1 b)

    The forward keeps the Q shifted solves of b; the backward runs msMINRES once more for the incoming gradient g (same shifts) and feeds
    the Q t (left, right) column pairs to ONE fused bilinear-derivative pass (``kv_grad2``); d/db = K^{-1/2} g is the weighted sum of the
    same solves."""

    @staticmethod
    def forward(ctx, x, lengthscale, outputscale, noise, rhs, spec, kparam=None):
        from . import backend as B
        from .functions import _prep

        n = x.shape[-2]
        xp = _prep(spec, x, lengthscale)
        wd = xp.dtype
        os_ = None if outputscale is None else outputscale.detach().reshape(-1)[:1].to(wd).contiguous()
        nz = noise.detach().reshape(-1)[:1].to(wd).contiguous()

        def matvec(vt):
            return B.kv(xp, xp, vt, scale=os_, dscale=nz, vd=vt, dvec=spec.dvec)

        rhs_t = B.to_probe_major(rhs, wd)
        from . import settings

        lo, hi = lanczos_eig_bounds(matvec, n, rhs_t.device, rhs_t.dtype)
        lmin, lmax = lo * 0.5, hi * 1.1
        weights, shifts = ciq_weights_shifts(lmin, lmax, settings.num_contour_quadrature.value())
        X, iters = msminres(matvec, rhs_t, shifts, n)
        w = weights.to(device=X.device, dtype=X.dtype)
        res = (w.reshape(-1, 1, 1) * X).sum(0)
        ctx.xp, ctx.n, ctx.matvec, ctx.shifts, ctx.w, ctx.X = xp, n, matvec, shifts, w, X
        ctx.kparam, ctx.has_os, ctx.x_dtype = kparam, outputscale is not None, x.dtype
        ctx.save_for_backward(lengthscale, outputscale if outputscale is not None else torch.empty(0), noise, rhs)
        ctx.info = dict(iterations=iters, lmin=lmin, lmax=lmax)
        return B.from_probe_major(res, n).to(rhs.dtype)

    @staticmethod
    def backward(ctx, grad_out):
        from . import backend as B
        from .functions import hyper_grads

        lengthscale, outputscale, noise, rhs = ctx.saved_tensors
        outputscale = outputscale if ctx.has_os else None
        xp, n, w = ctx.xp, ctx.n, ctx.w
        Q, t = ctx.X.shape[0], ctx.X.shape[1]
        g_t = B.to_probe_major(grad_out, xp.dtype)
        G, _ = msminres(ctx.matvec, g_t, ctx.shifts, n)                      # (K + s_q)^-1 g   [Q, t, ld]
        d_rhs = B.from_probe_major((w.reshape(-1, 1, 1) * G).sum(0), n).to(rhs.dtype) if ctx.needs_input_grad[4] else None
        left = (-(w.reshape(-1, 1, 1)) * G).reshape(Q * t, -1).contiguous()
        right = ctx.X.reshape(Q * t, -1).contiguous()
        kp = ctx.kparam if (ctx.kparam is not None and ctx.needs_input_grad[6]) else None
        d_x = d_par = None
        if ctx.needs_input_grad[0]:
            out = hyper_grads(xp, xp, lengthscale, outputscale, left, right, want_x1=True, want_x2=True, kparam=kp)
            d_ls, d_os = out[:2]
            d_x = (out[2] + out[3]).to(ctx.x_dtype)
        else:
            out = hyper_grads(xp, xp, lengthscale, outputscale, left, right, kparam=kp)
            d_ls, d_os = out[:2]
        if kp is not None:
            d_par = out[-1]
        d_noise = B.coldot(left, right, n).sum().reshape(noise.shape).to(noise.dtype)
        return d_x, d_ls, d_os, d_noise, d_rhs, None, d_par


def sqrt_inv_matmul(op, rhs: torch.Tensor, lhs: torch.Tensor | None = None):
    """``gpytorch.sqrt_inv_matmul(mat, rhs, lhs=None)`` (``gpytorch/__init__.py:252-278``): K^{-1/2} rhs, or
    (lhs K^{-1/2} rhs, diag-free inverse quadratic  lhs K^{-1} lhs^T summed over columns) when ``lhs`` is given.
    ``op``: any operator of this package (its ``_matmul`` is the matrix-free product).  Differentiable (``SqrtInvMatmulFn``: hyper-
    parameters, inputs, right-hand side) for the fused single-kernel operators when gradients are enabled; forward-only otherwise."""
    from . import backend as B

    n = op.shape[-1]
    squeeze = rhs.dim() == 1
    r = rhs.unsqueeze(-1) if squeeze else rhs
    from .operators import FusedKernelAddedDiagLinearOperator, FusedKernelLinearOperator

    fused = op if isinstance(op, FusedKernelAddedDiagLinearOperator) else None
    if isinstance(op, FusedKernelLinearOperator) and op.is_square:
        fused = op.add_jitter(0.0)
    if (lhs is None and fused is not None and torch.is_grad_enabled() and (fused.requires_grad or r.requires_grad)
            and fused.kernel_op.prepared()[0].fused):
        k = fused.kernel_op
        out = SqrtInvMatmulFn.apply(k.x1, k.lengthscale, k.outputscale, fused.noise, r, fused._spec(), k.spec.param)
        return out.squeeze(-1) if squeeze else out

    def matvec(vt):
        out = op._matmul(vt[:, :n].t().to(op.dtype))
        res = torch.zeros_like(vt)
        res[:, :n] = out.t().to(vt.dtype)
        return res

    with torch.no_grad():
        wd = torch.float64 if r.dtype == torch.float64 else torch.float32
        if lhs is None:
            sol_t, _ = contour_integral_quad(matvec, B.to_probe_major(r, wd), n)
            out = B.from_probe_major(sol_t, n).to(rhs.dtype)
            return out.squeeze(-1) if squeeze else out
        both = torch.cat([r, lhs.mT], dim=-1)
        sol_t, _ = contour_integral_quad(matvec, B.to_probe_major(both, wd), n)
        sol = B.from_probe_major(sol_t, n).to(rhs.dtype)
        k_rhs, k_lhs = sol[:, : r.shape[-1]], sol[:, r.shape[-1]:]
        return lhs @ k_rhs, k_lhs.pow(2).sum(dim=-2)


def sqrt_matmul(op, rhs: torch.Tensor):
    """K^{1/2} rhs (posterior sampling: mu + K^{1/2} eps)."""
    from . import backend as B

    n = op.shape[-1]
    squeeze = rhs.dim() == 1
    r = rhs.unsqueeze(-1) if squeeze else rhs

    def matvec(vt):
        out = op._matmul(vt[:, :n].t().to(op.dtype))
        res = torch.zeros_like(vt)
        res[:, :n] = out.t().to(vt.dtype)
        return res

    with torch.no_grad():
        wd = torch.float64 if r.dtype == torch.float64 else torch.float32
        sol_t, _ = contour_integral_quad(matvec, B.to_probe_major(r, wd), n, inverse=False)
    out = B.from_probe_major(sol_t, n).to(rhs.dtype)
    return out.squeeze(-1) if squeeze else out
