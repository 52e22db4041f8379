"""Autograd Functions that put the HIP path under torch autograd.

Mirror (third-party linear_operator):
  * ``functions/_inv_quad_logdet.py::InvQuadLogdet``  (forward: mBCG + SLQ; backward: ONE bilinear
    derivative with left = [K^-1 z c | -K^-1 y], right = [P^-1 z | K^-1 y])
  * ``functions/_matmul.py::Matmul`` / ``KernelLinearOperator._bilinear_derivative``
and the kernel-side backward of ``gpytorch/functions/rbf_covariance.py:26-29`` /
``matern_covariance.py:53-56``.  Gradients with respect to the INPUT LOCATIONS x -- which those dense Functions
refuse (``rbf_covariance.py:9-10``) and the KeOps precedent provides (``gpytorch/test/base_keops_test_case.py:105-132``)
-- come from the same fused pass (``kv_grad2.hpp``); outside its accuracy policy they are refused loudly, never silently zero.
"""
from __future__ import annotations

import torch

from . import backend as B
from .distributed import allreduce_sum_
from .bbmm import inv_quad_logdet_forward


class KernelSpec:
    """Non-tensor description of a stationary kernel operator (kind, centring shift, probe options)."""

    def __init__(self, kind: str, shift=None, dvec=None, param=None):
        self.kind = kind
        self.shift = shift
        self.dvec = dvec  # optional fixed (non-learnable) per-point noise diagonal, float32 [n] on the device
        self.param = param  # shape parameter of the covariance family as a (possibly learnable) tensor: RQ alpha; else None

    def with_dvec(self, dvec):
        return KernelSpec(self.kind, self.shift, dvec, self.param)

    def param_value(self):
        """The shape parameter as a Python float for the C ABI (one host read per evaluation), or None."""
        return None if self.param is None else float(self.param.detach().reshape(-1)[0])


def _prep(spec: "KernelSpec", x, lengthscale):
    return B.prep_points(spec.kind, x, lengthscale, spec.shift, spec.param_value())


def hyper_grads(xp1, xp2, lengthscale, outputscale, left_t, right_t, want_x1=False, want_x2=False, kparam=None):
    """d/d(lengthscale), d/d(outputscale) of  sum_c left[c]^T (outputscale * k(x1, x2)) right[c]  and, on request, its
    gradients with respect to the input locations x1 / x2 ([n, d] / [m, d], in the units of the ORIGINAL points).

    Returns (d_ls, d_os) or, with ``want_x1`` / ``want_x2``, (d_ls, d_os, d_x1, d_x2); with ``kparam`` (the learnable shape
    parameter tensor of the family: RQ alpha) the gradient with respect to it is appended as the LAST element."""
    wd = xp1.dtype
    ls = lengthscale.detach().to(wd).reshape(-1)
    iso = ls.numel() == 1
    gz1 = gz2 = None
    want_x = want_x1 or want_x2
    # beyond 16 dimensions the Gram-form derivative kernel exists for ONE lengthscale without input gradients (kv_grad2.hpp MODE 0: its work does not
    # grow with d); per-dimension sums take the direct-difference kernel, input gradients the row-block path
    per_dim_ok = xp1.d <= B.MAX_GRAD2_ARD_DIM or (iso and not want_x)
    if B.grad_gram_ok(xp1, xp2) and per_dim_ok:
        g, gz1 = B.kv_grad2(xp1, xp2, left_t, right_t, iso=iso, want_gz1=want_x1)
        if want_x2:  # the same kernel with the roles of the two clouds exchanged
            # gram_mode is not symmetric (it needs a compact sorted view of its FIRST cloud): the exchanged call has its own check
            if xp2 is not xp1 and not B.grad_gram_ok(xp2, xp1):
                raise RuntimeError(
                    "gradients with respect to the second input cloud need the Gram-form derivative kernel with the clouds exchanged, and "
                    "k(x2, x1) is outside its accuracy policy (backend.gram_mode(x2, x1) == 0: x2 too small or too wide for the block-centred expansion)"
                )
            _, gz2 = B.kv_grad2(xp2, xp1, right_t, left_t, iso=iso, want_gz1=True)
    elif want_x and xp1.fused and xp2.fused and not per_dim_ok:
        # 17 .. 32 dimensions with input gradients: dense row blocks (HIP generation + library GEMMs, float64 sums), both roles
        g, gz1 = B.kv_grad_generic(xp1, xp2, left_t, right_t, want_gz1=True)
        g = g.to(wd)
        gz1 = gz1.to(wd) if want_x1 else None
        if want_x2:
            gz2 = B.kv_grad_generic(xp2, xp1, right_t, left_t, want_gz1=True)[1].to(wd)
    else:
        if want_x:
            raise RuntimeError(
                "gradients with respect to the inputs need the Gram-form derivative kernel (float32, d <= 16, RBF / Matern "
                "3/2 / 5/2, max |x / lengthscale|^2 within the accuracy policy); this operator is outside it"
            )
        if xp1.fused and xp2.fused and xp1.kind != "rq":
            g = B.kv_grad(xp1, xp2, left_t, right_t, iso=iso)
        else:  # float64, d > 32, or a parametrised family outside the Gram-form accuracy policy
            g = B.kv_grad_generic(xp1, xp2, left_t, right_t).to(wd)
    d = xp1.d
    theta = 1.0 if outputscale is None else outputscale.detach().reshape(()).to(wd)
    gq = g[1 : 1 + d]
    if iso:
        d_ls = (theta * (-2.0) / ls * gq.sum()).reshape(lengthscale.shape)
    else:
        d_ls = (theta * (-2.0) / ls * gq).reshape(lengthscale.shape)
    d_os = None if outputscale is None else g[0].reshape(outputscale.shape)
    d_ls, d_os = d_ls.to(lengthscale.dtype), (None if d_os is None else d_os.to(outputscale.dtype))
    extra = ()
    if kparam is not None:
        # RQ: s = |dx|^2 / (2 alpha l^2)  ->  dK/dalpha = dk/dalpha|_s + dk/ds * (-s / alpha); the kernel delivers
        # g[1 + dp] = sum W dk/dalpha|_s and sum_q g[1 + q] = sum W dk/ds * s
        if g.numel() < 2 + xp1.dp:
            raise RuntimeError("the shape-parameter gradient needs the Gram-form derivative kernel")
        d_par = theta * (g[1 + xp1.dp] - gq.sum() / xp1.param)
        extra = (d_par.reshape(kparam.shape).to(kparam.dtype),)
    if not (want_x1 or want_x2):
        return (d_ls, d_os) + extra
    chain = theta * B.prep_coef_of(xp1) / ls  # dz/dx per dimension (1 or d values), times the outputscale
    return (d_ls, d_os, (None if gz1 is None else gz1 * chain), (None if gz2 is None else gz2 * chain)) + extra


class InvQuadLogdetFn(torch.autograd.Function):
    """(inv_quad[c], logdet) of K_hat = outputscale * k(x, x; lengthscale) + noise * I on the BBMM path."""

    @staticmethod
    def forward(ctx, x, lengthscale, outputscale, noise, rhs, spec: KernelSpec, opts: dict, kparam=None):
        n = x.shape[-2]
        xp = _prep(spec, x, lengthscale)
        ctx.kparam = kparam
        os_ = None if outputscale is None else outputscale.detach().reshape(-1)[:1].to(xp.dtype).contiguous()
        nz = noise.detach().reshape(-1)[:1].to(xp.dtype).contiguous()
        rhs_t = B.to_probe_major(rhs, xp.dtype)
        res = inv_quad_logdet_forward(
            xp, os_, nz, rhs_t,
            num_probes=opts.get("num_probes"), precond=opts.get("precond", "auto"), probes=opts.get("probes"),
            generator=opts.get("generator"), tolerance=opts.get("tolerance"), max_iter=opts.get("max_iter"),
            group=opts.get("group"), t_total=opts.get("t_total"), dvec=spec.dvec, row_group=opts.get("row_group"),
        )
        ctx.xp, ctx.res, ctx.n = xp, res, n
        ctx.x_dtype = x.dtype
        ctx.group = opts.get("group")
        ctx.t_total = opts.get("t_total") or res.zt.shape[0]
        ctx.save_for_backward(lengthscale, outputscale if outputscale is not None else torch.empty(0), noise, rhs)
        ctx.has_os = outputscale is not None
        opts["_last_info"] = res.info
        return res.inv_quad.to(rhs.dtype), res.logdet.to(rhs.dtype)

    @staticmethod
    def backward(ctx, g_iq, g_ld):
        lengthscale, outputscale, noise, rhs = ctx.saved_tensors
        outputscale = outputscale if ctx.has_os else None
        res, xp, n = ctx.res, ctx.xp, ctx.n
        t = res.zt.shape[0]
        c = res.solves_t.shape[0] - t
        g_iq = g_iq.to(xp.dtype).reshape(c, 1)
        g_ld = g_ld.to(xp.dtype).reshape(())
        s_z = res.solves_t[:t] * res.znorm.unsqueeze(-1)
        s_y = res.solves_t[t:]
        zr = res.zt * res.znorm.unsqueeze(-1)
        if res.precond is not None:
            zr = res.precond.apply_(zr, torch.zeros_like(zr))
        if res.owns_rhs:
            left = torch.cat([s_z * (g_ld / ctx.t_total), -s_y * g_iq], dim=0).contiguous()
            right = torch.cat([zr, s_y], dim=0).contiguous()
        else:  # sharded: the rhs block is differentiated by its owner only; the all-reduce below adds it once
            left = (s_z * (g_ld / ctx.t_total)).contiguous()
            right = zr.contiguous()
        d_x = d_par = None
        kp = ctx.kparam if (ctx.kparam is not None and ctx.needs_input_grad[7]) else None
        if ctx.needs_input_grad[0]:
            out = hyper_grads(xp, xp, lengthscale, outputscale, left, right, want_x1=True, want_x2=True, kparam=kp)
            d_ls, d_os, gx1, gx2 = out[:4]
            d_x = (gx1 + gx2).to(ctx.x_dtype)
        else:
            out = hyper_grads(xp, xp, lengthscale, outputscale, left, right, kparam=kp)
            d_ls, d_os = out[:2]
        if kp is not None:
            d_par = out[-1]
            if ctx.group is not None:
                allreduce_sum_(d_par, ctx.group)
        d_noise = B.coldot(left, right, n).sum().reshape(noise.shape).to(noise.dtype)
        if ctx.group is not None:
            pack = torch.cat([d_ls.reshape(-1).to(xp.dtype), d_noise.reshape(-1).to(xp.dtype)] + ([d_os.reshape(-1).to(xp.dtype)] if d_os is not None else []))
            allreduce_sum_(pack, ctx.group)
            k = d_ls.numel()
            d_ls = pack[:k].reshape(d_ls.shape).to(d_ls.dtype)
            d_noise = pack[k : k + 1].reshape(d_noise.shape).to(d_noise.dtype)
            if d_os is not None:
                d_os = pack[k + 1 :].reshape(d_os.shape).to(d_os.dtype)
        d_rhs = None
        if ctx.needs_input_grad[4]:
            d_rhs = (2.0 * B.from_probe_major(s_y, n) * g_iq.reshape(1, c)).to(rhs.dtype)
        if d_x is not None and ctx.group is not None:
            allreduce_sum_(d_x, ctx.group)
        return d_x, d_ls, d_os, d_noise, d_rhs, None, None, d_par


class CholeskyInvQuadLogdetFn(torch.autograd.Function):
    """Small-n branch (n <= max_cholesky_size or fast_computations.log_prob off): K_hat is formed by
    the HIP dense kernel, factorised by rocSOLVER through torch; exact inv_quad / logdet."""

    @staticmethod
    def forward(ctx, x, lengthscale, outputscale, noise, rhs, spec: KernelSpec, kparam=None):
        n = x.shape[-2]
        xp = _prep(spec, x, lengthscale)
        ctx.kparam = kparam
        os_ = None if outputscale is None else outputscale.detach().reshape(-1)[:1].to(xp.dtype).contiguous()
        K = B.kernel_dense(xp, xp, os_).to(torch.float64)
        K.diagonal().add_(noise.detach().reshape(()).to(torch.float64))
        if spec.dvec is not None:
            K.diagonal().add_(spec.dvec[:n].to(torch.float64))
        from .operators import psd_safe_cholesky   # (operators imports this module)

        Lc = psd_safe_cholesky(K, model_dtype=rhs.dtype)
        sol = torch.cholesky_solve(rhs.detach().to(torch.float64), Lc)
        inv_quad = (sol * rhs.detach().to(torch.float64)).sum(-2)
        logdet = 2.0 * Lc.diagonal().log().sum()
        ctx.xp, ctx.n = xp, n
        ctx.save_for_backward(lengthscale, outputscale if outputscale is not None else torch.empty(0), noise, rhs, Lc, sol)
        ctx.has_os = outputscale is not None
        return inv_quad.to(rhs.dtype), logdet.to(rhs.dtype)

    @staticmethod
    def backward(ctx, g_iq, g_ld):
        lengthscale, outputscale, noise, rhs, Lc, sol = ctx.saved_tensors
        outputscale = outputscale if ctx.has_os else None
        n, xp = ctx.n, ctx.xp
        c = sol.shape[-1]
        kinv = torch.cholesky_inverse(Lc)
        # d logdet = tr(K^-1 dK);  d inv_quad = -sol^T dK sol
        left = torch.cat([kinv * g_ld.to(torch.float64), -(sol * g_iq.to(torch.float64).reshape(1, c)).t()], dim=0)
        right = torch.cat([torch.eye(n, device=sol.device, dtype=torch.float64), sol.t()], dim=0)
        ld = B.round_up(n, 4)
        lt = torch.zeros(n + c, ld, device=sol.device, dtype=xp.dtype)
        rt = torch.zeros(n + c, ld, device=sol.device, dtype=xp.dtype)
        lt[:, :n] = left
        rt[:, :n] = right
        d_x = d_par = None
        kp = ctx.kparam if (ctx.kparam is not None and ctx.needs_input_grad[6]) else None
        if ctx.needs_input_grad[0]:
            out = hyper_grads(xp, xp, lengthscale, outputscale, lt, rt, want_x1=True, want_x2=True, kparam=kp)
            d_ls, d_os, gx1, gx2 = out[:4]
            d_x = gx1 + gx2
        else:
            out = hyper_grads(xp, xp, lengthscale, outputscale, lt, rt, kparam=kp)
            d_ls, d_os = out[:2]
        if kp is not None:
            d_par = out[-1]
        d_noise = (left * right).sum().reshape(noise.shape).to(noise.dtype)
        d_rhs = (2.0 * sol * g_iq.to(torch.float64).reshape(1, c)).to(rhs.dtype) if ctx.needs_input_grad[4] else None
        return d_x, d_ls, d_os, d_noise, d_rhs, None, d_par


class KernelMatmulFn(torch.autograd.Function):
    """(outputscale * k(x1, x2)) @ rhs (+ noise * rhs when square and noise is given)."""

    @staticmethod
    def forward(ctx, x1, x2, lengthscale, outputscale, noise, rhs, spec: KernelSpec, kparam=None):
        ctx.same_x = x2 is x1
        ctx.kparam = kparam
        xp1 = _prep(spec, x1, lengthscale)
        xp2 = xp1 if x2 is x1 else _prep(spec, x2.to(x1.dtype), lengthscale)
        os_ = None if outputscale is None else outputscale.detach().reshape(-1)[:1].to(xp1.dtype).contiguous()
        nz = None if noise is None else noise.detach().reshape(-1)[:1].to(xp1.dtype).contiguous()
        vt = B.to_probe_major(rhs, xp1.dtype)
        out_t = B.kv(xp1, xp2, vt, scale=os_, dscale=nz, vd=vt if nz is not None else None,
                     dvec=spec.dvec if nz is not None else None)
        ctx.xp1, ctx.xp2, ctx.dvec = xp1, xp2, (spec.dvec if nz is not None else None)
        ctx.save_for_backward(lengthscale, outputscale if outputscale is not None else torch.empty(0),
                              noise if noise is not None else torch.empty(0), rhs)
        ctx.has_os, ctx.has_noise = outputscale is not None, noise is not None
        return B.from_probe_major(out_t, xp1.n).to(rhs.dtype)

    @staticmethod
    def backward(ctx, g):
        lengthscale, outputscale, noise, rhs = ctx.saved_tensors
        outputscale = outputscale if ctx.has_os else None
        noise = noise if ctx.has_noise else None
        wd = ctx.xp1.dtype
        gt = B.to_probe_major(g, wd)
        rt = B.to_probe_major(rhs, wd)
        d_ls = d_os = d_noise = d_rhs = d_x1 = d_x2 = d_par = None
        kp = ctx.kparam if (ctx.kparam is not None and ctx.needs_input_grad[7]) else None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            out = hyper_grads(ctx.xp1, ctx.xp2, lengthscale, outputscale, gt, rt, want_x1=ctx.needs_input_grad[0],
                              want_x2=ctx.needs_input_grad[1], kparam=kp)
            d_ls, d_os, d_x1, d_x2 = out[:4]
        elif ctx.needs_input_grad[2] or ctx.needs_input_grad[3] or kp is not None:
            out = hyper_grads(ctx.xp1, ctx.xp2, lengthscale, outputscale, gt, rt, kparam=kp)
            d_ls, d_os = out[:2]
        if kp is not None:
            d_par = out[-1]
        if noise is not None and ctx.needs_input_grad[4]:
            d_noise = (g * rhs).sum().reshape(noise.shape).to(noise.dtype)
        if ctx.needs_input_grad[5]:
            os_ = None if outputscale is None else outputscale.detach().reshape(-1)[:1].to(wd).contiguous()
            nz = None if noise is None else noise.detach().reshape(-1)[:1].to(wd).contiguous()
            out_t = B.kv(ctx.xp2, ctx.xp1, gt, scale=os_, dscale=nz, vd=gt if nz is not None else None, dvec=ctx.dvec)
            d_rhs = B.from_probe_major(out_t, ctx.xp2.n).to(rhs.dtype)
        return d_x1, d_x2, d_ls, d_os, d_noise, d_rhs, None, d_par


class KernelDenseFn(torch.autograd.Function):
    """``to_dense()`` of outputscale * k(x1, x2) with hyper-parameter gradients (the eager branch of
    ``exact_prediction``, ``exact_prediction_strategies.py:331-369``, when n + m <= max_eager_kernel_size).
    Backward: sum_ij G_ij dK_ij/dtheta = the fused bilinear derivative with left = I, right = G."""

    @staticmethod
    def forward(ctx, x1, x2, lengthscale, outputscale, spec: KernelSpec, kparam=None):
        ctx.kparam = kparam
        xp1 = _prep(spec, x1, lengthscale)
        xp2 = xp1 if x2 is x1 else _prep(spec, x2.to(x1.dtype), lengthscale)
        os_ = None if outputscale is None else outputscale.detach().reshape(-1)[:1].to(xp1.dtype).contiguous()
        ctx.xp1, ctx.xp2 = xp1, xp2
        ctx.save_for_backward(lengthscale, outputscale if outputscale is not None else torch.empty(0))
        ctx.has_os = outputscale is not None
        return B.kernel_dense(xp1, xp2, os_).to(x1.dtype)

    @staticmethod
    def backward(ctx, g):
        lengthscale, outputscale = ctx.saved_tensors
        outputscale = outputscale if ctx.has_os else None
        xp1, xp2 = ctx.xp1, ctx.xp2
        n, m = xp1.n, xp2.n
        wd = xp1.dtype
        lt = torch.zeros(n, B.round_up(n, 4), device=g.device, dtype=wd)
        lt[:, :n] = torch.eye(n, device=g.device, dtype=wd)
        rt = torch.zeros(n, B.round_up(m, 4), device=g.device, dtype=wd)
        rt[:, :m] = g.to(wd)
        kp = ctx.kparam if (ctx.kparam is not None and ctx.needs_input_grad[5]) else None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            out = hyper_grads(xp1, xp2, lengthscale, outputscale, lt, rt, want_x1=ctx.needs_input_grad[0],
                              want_x2=ctx.needs_input_grad[1], kparam=kp)
            return out[2], out[3], out[0], out[1], None, (out[-1] if kp is not None else None)
        out = hyper_grads(xp1, xp2, lengthscale, outputscale, lt, rt, kparam=kp)
        return None, None, out[0], out[1], None, (out[-1] if kp is not None else None)


class SolveFn(torch.autograd.Function):
    """K_hat^-1 rhs by preconditioned mBCG with the reference's ``Solve`` backward (linear_operator functions/_solve.py;
    exercised by test/lazy/test_lazy_evaluated_kernel_tensor.py:69-113): with X = K_hat^-1 B and Y = K_hat^-1 G,
    d/dB = Y and d/dtheta = -sum_c Y_c^T (dK_hat/dtheta) X_c -- ONE more mBCG solve and ONE fused bilinear derivative."""

    @staticmethod
    def forward(ctx, x, lengthscale, outputscale, noise, rhs, spec: KernelSpec, precond, tolerance, kparam=None):
        from .linear_cg import linear_cg

        ctx.kparam = kparam
        xp = _prep(spec, x, lengthscale)
        os_ = None if outputscale is None else outputscale.detach().reshape(-1)[:1].to(xp.dtype).contiguous()
        nz = noise.detach().reshape(-1)[:1].to(xp.dtype).contiguous()
        sol_t, info = linear_cg(xp, os_, nz, B.to_probe_major(rhs.detach(), xp.dtype), n_tridiag=0, tolerance=tolerance,
                                preconditioner=precond, dvec=spec.dvec)
        ctx.xp, ctx.os_, ctx.nz, ctx.sol_t, ctx.spec, ctx.precond, ctx.tol = xp, os_, nz, sol_t, spec, precond, tolerance
        ctx.info = info
        ctx.save_for_backward(lengthscale, outputscale if outputscale is not None else torch.empty(0), noise)
        ctx.has_os = outputscale is not None
        ctx.rhs_dtype = rhs.dtype
        return B.from_probe_major(sol_t, xp.n).to(rhs.dtype)

    @staticmethod
    def backward(ctx, g):
        from .linear_cg import linear_cg

        lengthscale, outputscale, noise = ctx.saved_tensors
        outputscale = outputscale if ctx.has_os else None
        xp = ctx.xp
        y_t, _ = linear_cg(xp, ctx.os_, ctx.nz, B.to_probe_major(g, xp.dtype), n_tridiag=0, tolerance=ctx.tol,
                           preconditioner=ctx.precond, dvec=ctx.spec.dvec)
        left = -y_t
        d_ls = d_os = d_noise = d_rhs = d_x = d_par = None
        kp = ctx.kparam if (ctx.kparam is not None and ctx.needs_input_grad[8]) else None
        if ctx.needs_input_grad[0]:
            out = hyper_grads(xp, xp, lengthscale, outputscale, left.contiguous(), ctx.sol_t, want_x1=True, want_x2=True, kparam=kp)
            d_ls, d_os, gx1, gx2 = out[:4]
            d_x = gx1 + gx2
        elif ctx.needs_input_grad[1] or ctx.needs_input_grad[2] or kp is not None:
            out = hyper_grads(xp, xp, lengthscale, outputscale, left.contiguous(), ctx.sol_t, kparam=kp)
            d_ls, d_os = out[:2]
        if kp is not None:
            d_par = out[-1]
        if ctx.needs_input_grad[3]:
            d_noise = B.coldot(left.contiguous(), ctx.sol_t, xp.n).sum().reshape(noise.shape).to(noise.dtype)
        if ctx.needs_input_grad[4]:
            d_rhs = B.from_probe_major(y_t, xp.n).to(ctx.rhs_dtype)
        return d_x, d_ls, d_os, d_noise, d_rhs, None, None, None, d_par
