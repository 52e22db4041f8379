"""Sums of fused kernel operators: ``AdditiveKernel`` (``gpytorch/kernels/kernel.py:592-632``) on the matrix-free path.

``K = sum_i theta_i k_i(x, x)`` stays a list of :class:`~gpytorch_amd.operators.FusedKernelLinearOperator` members; products,
mBCG solves, SLQ log-determinants and Lanczos decompositions use the members' fused K*V launches summed per product (the
``kv_partials`` hook of :func:`gpytorch_amd.linear_cg.linear_cg`), and the A.6 backward runs one fused bilinear derivative per
member.  Preconditioner: pivoted Cholesky on rows of the SUMMED kernel (``sum_preconditioner``; ``bbmm.pivoted_cholesky_rows``).
"""
from __future__ import annotations

import torch

from . import backend as B
from . import settings
from .bbmm import allreduce_grads_, backward_vectors, build_preconditioner_rows, inv_quad_logdet_forward, structured_opts
from .functions import _prep, hyper_grads
from .linear_cg import linear_cg
from .operators import (
    DiagLinearOperator,
    FusedKernelLinearOperator,
    LinearOperator,
    RootLinearOperator,
    ZeroLinearOperator,
    check_root_method,
    lanczos_vectors,
    psd_safe_cholesky,
    split_diag,
)


class SumFusedLinearOperator(LinearOperator):
    """sum_i members[i], every member a (scaled) fused kernel operator over the same pair of point clouds."""

    def __init__(self, ops):
        self.ops = list(ops)

    @classmethod
    def of(cls, terms):
        flat = []
        for t in terms:
            flat.extend(t.ops if isinstance(t, SumFusedLinearOperator) else [t])
        if len(flat) == 1:
            return flat[0]
        if not all(isinstance(t, FusedKernelLinearOperator) for t in flat):
            from .operators import SumLinearOperator

            return SumLinearOperator(*flat)  # generic (dense-capable) sum
        return cls(flat)

    dtype = property(lambda self: self.ops[0].dtype)
    device = property(lambda self: self.ops[0].device)

    @property
    def requires_grad(self):
        return any(o.requires_grad for o in self.ops)

    def _size(self):
        return self.ops[0].shape

    def _matmul(self, rhs):
        out = self.ops[0]._matmul(rhs)
        for o in self.ops[1:]:
            out = out + o._matmul(rhs)
        return out

    def _transpose_nonbatch(self):
        return SumFusedLinearOperator([o._transpose_nonbatch() for o in self.ops])

    def _mul_constant(self, c):
        return SumFusedLinearOperator([o._mul_constant(c) for o in self.ops])

    def to_dense(self):
        return sum((o.to_dense() for o in self.ops[1:]), self.ops[0].to_dense())

    def diagonal(self, offset=0, dim1=-2, dim2=-1):
        return sum((o.diagonal() for o in self.ops[1:]), self.ops[0].diagonal())

    def __getitem__(self, index):
        parts = [o[index] for o in self.ops]
        if all(isinstance(p, FusedKernelLinearOperator) for p in parts):
            return SumFusedLinearOperator(parts)
        return sum(parts[1:], parts[0])

    def detach(self):
        return SumFusedLinearOperator([o.detach() for o in self.ops])

    def __add__(self, other):
        if isinstance(other, ZeroLinearOperator):
            return self
        if isinstance(other, (FusedKernelLinearOperator, SumFusedLinearOperator)):
            return SumFusedLinearOperator.of([self, other])
        if isinstance(other, DiagLinearOperator) and self.is_square and not other.batch_shape:
            noise, vec = split_diag(other, self.device, self.dtype)
            return SumFusedAddedDiagLinearOperator(self, noise, noise_vec=vec)
        return super().__add__(other)

    __radd__ = __add__


class SumFusedAddedDiagLinearOperator(LinearOperator):
    """(sum_i theta_i k_i(x, x)) + noise I (+ diag(noise_vec)): the operator the MLL and the prediction caches solve with."""

    def __init__(self, kernel_sum: SumFusedLinearOperator, noise: torch.Tensor, noise_vec=None, bbmm_opts=None):
        self.kernel_sum = kernel_sum
        self.noise = noise.reshape(-1)[:1]
        self.noise_vec = noise_vec
        self.bbmm_opts = {} if bbmm_opts is None else bbmm_opts

    dtype = property(lambda self: self.kernel_sum.dtype)
    device = property(lambda self: self.kernel_sum.device)

    @property
    def requires_grad(self):
        return self.kernel_sum.requires_grad or self.noise.requires_grad

    def _size(self):
        return self.kernel_sum._size()

    def _diag_total(self):
        n = self.shape[-1]
        d = self.noise.reshape(()).expand(n)
        return d if self.noise_vec is None else d + self.noise_vec

    def _matmul(self, rhs):
        return self.kernel_sum._matmul(rhs) + self._diag_total().unsqueeze(-1) * rhs

    def _transpose_nonbatch(self):
        return self

    def diagonal(self, offset=0, dim1=-2, dim2=-1):
        return self.kernel_sum.diagonal() + self._diag_total()

    def to_dense(self):
        return self.kernel_sum.to_dense() + torch.diag(self._diag_total().to(self.dtype))

    def detach(self):
        return SumFusedAddedDiagLinearOperator(self.kernel_sum.detach(), self.noise.detach(), self.noise_vec, self.bbmm_opts)

    def __add__(self, other):
        if isinstance(other, DiagLinearOperator) and not other.batch_shape:
            noise, vec = split_diag(other, self.device, self.dtype)
            nv = self.noise_vec if vec is None else (vec if self.noise_vec is None else self.noise_vec + vec)
            return SumFusedAddedDiagLinearOperator(self.kernel_sum, self.noise + noise, nv, self.bbmm_opts)
        return super().__add__(other)

    def restrict(self, idx):
        nv = None if self.noise_vec is None else self.noise_vec[idx]
        return SumFusedAddedDiagLinearOperator(self.kernel_sum[idx, idx], self.noise, nv, self.bbmm_opts)

    # ---- matrix-free machinery shared by the three entry points
    def _use_cholesky(self, flag) -> bool:
        return flag.off() or self.shape[-1] <= settings.max_cholesky_size.value()

    def _prepared(self):
        return [(o.prepared()[0], o._os()) for o in self.kernel_sum.ops]

    def _dvec(self, wd):
        if self.noise_vec is None:
            return None
        n = self.shape[-1]
        dv = torch.zeros(B.round_up(n, 4), device=self.device, dtype=wd)
        dv[:n] = self.noise_vec.detach().to(wd)
        return dv

    def inv_quad_logdet(self, inv_quad_rhs=None, logdet=False, reduce_inv_quad=True):
        n = self.shape[-1]
        if inv_quad_rhs is None:
            inv_quad_rhs = torch.zeros(n, 0, device=self.device, dtype=self.dtype)
        rhs = inv_quad_rhs.unsqueeze(-1) if inv_quad_rhs.dim() == 1 else inv_quad_rhs
        if self._use_cholesky(settings.fast_computations.log_prob):
            # small problems: K_hat is formed from the members' differentiable dense kernels; torch differentiates the rest
            Kh = self.to_dense().to(torch.float64)
            Lc = psd_safe_cholesky(Kh, model_dtype=self.dtype)
            sol = torch.cholesky_solve(rhs.to(torch.float64), Lc)
            iq = (sol * rhs.to(torch.float64)).sum(-2).to(rhs.dtype)
            ld = (2.0 * Lc.diagonal().log().sum()).to(rhs.dtype)
        else:
            drop = rhs.shape[-1] == 0
            if drop:
                rhs = torch.zeros(n, 1, device=self.device, dtype=self.dtype)
            flat = []
            empty = torch.empty(0, device=self.device)
            for o in self.kernel_sum.ops:
                flat += [o.x1, o.lengthscale, o.outputscale if o.outputscale is not None else empty,
                         o.spec.param if o.spec.param is not None else empty]
            specs = [o.spec for o in self.kernel_sum.ops]
            iq, ld = SumInvQuadLogdetFn.apply(self.noise, rhs, specs, self._dvec(B.work_dtype(self.kernel_sum.ops[0].x1)), self.bbmm_opts, *flat)
            if drop:
                iq = iq[:0]
        if reduce_inv_quad:
            iq = iq.sum(-1)
        return iq, (ld if logdet else None)

    def inv_quad(self, inv_quad_rhs, reduce_inv_quad=True):
        return self.inv_quad_logdet(inv_quad_rhs, False, reduce_inv_quad)[0]

    def logdet(self):
        return self.inv_quad_logdet(None, True)[1]

    def _partials(self):
        prepared = self._prepared()

        def partials(dt):
            out = None
            for xp, os_ in prepared:
                q = B.kv(xp, xp, dt, scale=os_)
                out = q.clone() if out is None else out.add_(q)
            return out, 1, out.stride(0)

        return partials, prepared[0][0].dtype

    # ---- float64 product on the prepared points of the float32 path (mixed-precision corrections: settings.rhs_refinement; round 6) ----
    def float64_product_available(self) -> bool:
        """Every member has a fused float64 product (float32 model, d <= 16: ``csrc/kv_f64.hpp``)."""
        return all(xp.fused and xp.dp <= B.FUSED_F64_MAX_DP for xp, _ in self._prepared())

    def _matvec64(self):
        """a64 [c, ld] (probe-major, float64) -> (sum_i theta_i K_i + sigma^2 I + D) a in float64: one fused float64 product per member
        (``bbmm.matvec64``), the diagonal once."""
        from .bbmm import matvec64

        mvs = [matvec64(xp, os_, None) for xp, os_ in self._prepared()]
        nz = self.noise.detach().reshape(()).to(torch.float64)
        dv = self._dvec(torch.float64)

        def mv(a64):
            out = mvs[0](a64)
            for m_ in mvs[1:]:
                out = out + m_(a64)
            out = out + nz * a64
            return out if dv is None else out + dv.unsqueeze(0) * a64

        return mv

    def matmul_float64(self, rhs: torch.Tensor) -> torch.Tensor:
        """K_hat @ rhs ([n, c] -> [n, c]) in float64 on the prepared points of the float32 path (``bbmm.variational_inv_quad``)."""
        if not self.float64_product_available():
            return None
        return B.from_probe_major(self._matvec64()(B.to_probe_major(rhs.detach(), torch.float64)), self.shape[-1])

    def solve(self, rhs, lhs=None):
        squeeze = rhs.dim() == 1
        r = rhs.unsqueeze(-1) if squeeze else rhs
        if self._use_cholesky(settings.fast_computations.solves):
            sol = torch.cholesky_solve(r.detach().to(torch.float64), psd_safe_cholesky(self.to_dense().detach().to(torch.float64), model_dtype=self.dtype)).to(rhs.dtype)
        else:
            partials, wd = self._partials()
            nz = self.noise.detach().reshape(-1)[:1].to(wd).contiguous()
            dv = self._dvec(wd)
            if not hasattr(self, "_precond_cache"):
                self._precond_cache = sum_preconditioner(self._prepared(), nz, dv, self.shape[-1])

            def cg32(rt):
                return linear_cg(None, None, nz, rt, n_tridiag=0, tolerance=settings.cg_tolerance.value(),
                                 kv_partials=partials, dvec=dv, nvec=self.shape[-1], preconditioner=self._precond_cache)

            rhs_t = B.to_probe_major(r.detach(), wd)
            sol_t, _ = cg32(rhs_t)
            if settings.rhs_refinement.on() and sol_t.dtype == torch.float32 and self.float64_product_available():
                # mixed-precision refinement of the float32 solves on the sum operator too (round 6): float64 residual through the members' fused
                # float64 products, one more float32 solve of it (bbmm.refine_with_) -- as the single-kernel and Kronecker operators do
                from .bbmm import refine_with_

                def solve32(res):
                    d_, inf = cg32(res)
                    return d_, inf.iterations

                refine_with_(rhs_t.to(torch.float64), sol_t, self._matvec64(), solve32, settings.rhs_refinement.steps)
            sol = B.from_probe_major(sol_t, self.shape[-1]).to(rhs.dtype)
        if lhs is not None:
            sol = lhs @ sol
        return sol.squeeze(-1) if squeeze else sol

    def root_inv_decomposition(self, initial_vectors=None, test_vectors=None, method=None):
        from .lanczos import root_inv_decomposition

        method = check_root_method(method, inverse=True)
        if method in ("cholesky", "symeig") or (method is None and self._use_cholesky(settings.fast_computations.covar_root_decomposition)):
            return super().root_inv_decomposition(method=method)      # dense factorisations of a small operator (base class)
        n = self.shape[-1]
        partials, wd = self._partials()
        nz = self.noise.detach().reshape(()).to(wd)
        dv = self._dvec(wd)

        def mv(q_row):
            out, _, _ = partials(q_row)
            out = out + nz * q_row
            return out if dv is None else out + dv.unsqueeze(0) * q_row

        init_t, test_t = lanczos_vectors(initial_vectors, test_vectors, n, wd)
        rt = root_inv_decomposition(None, None, None, matvec=mv, nvec=n, device=self.device, generator=self.bbmm_opts.get("generator"),
                                    init_vec_t=init_t, test_vec_t=test_t, dtype=wd)
        return RootLinearOperator(B.from_probe_major(rt, n).to(self.dtype))


def sum_preconditioner(prepared, noise, dvec, n, rank=None, tol=None, min_size=None):
    """Pivoted-Cholesky preconditioner of (sum_i theta_i k_i) + noise I (+ diag(dvec)): a row of the sum is the sum of the members'
    ``gpamd_kernel_rows_f32`` rows (the reference preconditions a SumLinearOperator + diagonal the same way, through its generic
    ``pivoted_cholesky``; ``gpytorch/kernels/kernel.py:592-632`` + ``settings.py:6-31``)."""
    if not all(xp.fused for xp, _ in prepared):
        return None
    wd = prepared[0][0].dtype

    def row_fn(p):
        out = None
        for xp, os_ in prepared:
            r = B.kernel_rows(xp, p, xp, os_).reshape(-1)
            out = r if out is None else out + r
        return out

    kdiag = None
    for xp, os_ in prepared:
        dg = B.kernel_diag(xp, xp, os_)
        kdiag = dg if kdiag is None else kdiag + dg
    nz = noise.detach().reshape(-1)[:1].to(wd)
    if dvec is None:
        return build_preconditioner_rows(row_fn, kdiag, nz, False, rank, tol, min_size)
    return build_preconditioner_rows(row_fn, kdiag, nz + dvec[:n].to(wd), True, rank, tol, min_size)


class SumInvQuadLogdetFn(torch.autograd.Function):
    """(inv_quad[c], logdet) of sum_i theta_i k_i(x_i, x_i) + noise I by preconditioned mBCG + SLQ (``bbmm.inv_quad_logdet_forward`` on the
    summed product; probe columns shardable over ``opts["group"]``); backward: one fused bilinear derivative per member with the shared
    left / right vectors of A.6.  ``flat``: (x, lengthscale, outputscale-or-empty, shape-parameter-or-empty) per member -- RQ members
    carry their alpha as an explicit argument all the way into the kernels (C ABI version 2), so two RQ members with different alpha
    interleave freely."""

    @staticmethod
    def forward(ctx, noise, rhs, specs, dvec, opts, *flat):
        nm = len(specs)
        members = [(flat[4 * i], flat[4 * i + 1], flat[4 * i + 2] if flat[4 * i + 2].numel() else None,
                    flat[4 * i + 3] if flat[4 * i + 3].numel() else None) for i in range(nm)]
        n = members[0][0].shape[-2]
        dev = rhs.device
        prepared = []
        for (x, ls, os_, _), spec in zip(members, specs):
            xp = _prep(spec, x, ls)
            prepared.append((xp, None if os_ is None else os_.detach().reshape(-1)[:1].to(xp.dtype).contiguous()))
        wd = prepared[0][0].dtype
        opts_in = opts
        opts = structured_opts(opts, dev)

        def partials(dt):
            out = None
            for xp, os_ in prepared:
                q = B.kv(xp, xp, dt, scale=os_)
                out = q.clone() if out is None else out.add_(q)
            return out, 1, out.stride(0)

        nz = noise.detach().reshape(-1)[:1].to(wd).contiguous()
        pre = opts.get("precond", "auto")
        if pre == "auto":
            pre = sum_preconditioner(prepared, nz, dvec, n)
        res = inv_quad_logdet_forward(
            None, None, nz, B.to_probe_major(rhs, wd), num_probes=opts.get("num_probes"), precond=pre, probes=opts.get("probes"),
            generator=opts.get("generator"), tolerance=opts.get("tolerance"), max_iter=opts.get("max_iter"), group=opts.get("group"),
            t_total=opts.get("t_total"), dvec=dvec, kv_partials=partials, nvec=n,
        )
        ctx.prepared, ctx.n, ctx.res = prepared, n, res
        ctx.members = members
        ctx.group = opts.get("group")
        ctx.t_total = opts.get("t_total") or res.zt.shape[0]
        ctx.save_for_backward(noise, rhs)
        opts_in["_last_info"] = res.info
        return res.inv_quad.to(rhs.dtype), res.logdet.to(rhs.dtype)

    @staticmethod
    def backward(ctx, g_iq, g_ld):
        noise, rhs = ctx.saved_tensors
        n, res = ctx.n, ctx.res
        left, right, s_y = backward_vectors(res, g_iq, g_ld, ctx.t_total)
        c = s_y.shape[0]
        wd = left.dtype
        grads = []
        for i, ((x, ls, os_, kpar), (xp, _)) in enumerate(zip(ctx.members, ctx.prepared)):
            need_x = ctx.needs_input_grad[5 + 4 * i]
            kp = kpar if (kpar is not None and ctx.needs_input_grad[5 + 4 * i + 3]) else None
            out = hyper_grads(xp, xp, ls, os_, left, right, want_x1=need_x, want_x2=need_x, kparam=kp)
            d_ls, d_os = out[:2]
            d_x = (out[2] + out[3]).to(x.dtype) if need_x else None
            d_par = out[-1] if kp is not None else None
            grads += [d_x, d_ls if ls.requires_grad else None, d_os, d_par]
        d_noise = B.coldot(left, right, n).sum().reshape(noise.shape).to(noise.dtype)
        allreduce_grads_([d_noise] + grads, ctx.group)
        d_rhs = (2.0 * B.from_probe_major(s_y, n) * g_iq.to(wd).reshape(1, c)).to(rhs.dtype) if ctx.needs_input_grad[1] else None
        return (d_noise, d_rhs, None, None, None, *grads)
