"""Hadamard multitask GPs on the fused path: ``K[i, j] = k(x_i, x_j) * K_TT[task_i, task_j]``.

Mirrors ``IndexKernel.forward`` (``gpytorch/kernels/index_kernel.py:101-112``: the task covariance looked up at the task index of
every point) and ``covar_x.mul(covar_i)`` in ``test/examples/test_hadamard_multitask_gp_regression.py:33-54``.  The reference
forms the elementwise product of an n x n kernel matrix with an interpolated task matrix; here K is never formed:

    (K o B)[i, :] v  =  sum_tau' K_TT[task_i, tau'] * ( k(x_i, X) (v o 1[task = tau']) )

i.e. ONE fused K*V launch with T * t columns (every right-hand side masked to one task) followed by an O(n T t) gather -- the
same "T*t columns" trick as the Kronecker operator of :mod:`gpytorch_amd.multitask`.  Solves, SLQ log-determinants and Lanczos
decompositions run on it; the A.6 backward is one fused bilinear derivative (T * t masked columns) for the data-kernel
hyper-parameters and one fused K*V for dK_TT.
"""
from __future__ import annotations

import torch

from . import backend as B
from . import settings
from .bbmm import allreduce_grads_, backward_vectors, build_preconditioner_rows, inv_quad_logdet_forward, structured_opts
from .functions import KernelSpec, _prep, hyper_grads
from .lanczos import root_inv_decomposition
from .linear_cg import linear_cg
from .operators import (DiagLinearOperator, FusedKernelLinearOperator, LinearOperator, RootLinearOperator, check_root_method, lanczos_vectors,
                        psd_safe_cholesky, split_diag)


class IndexedTaskCovar(LinearOperator):
    """K_TT[i1, i2]: what ``IndexKernel.forward(i1, i2)`` returns (an ``InterpolatedLinearOperator`` in the reference)."""

    def __init__(self, ktt: torch.Tensor, i1: torch.Tensor, i2: torch.Tensor):
        self.ktt = ktt
        self.i1, self.i2 = i1.reshape(-1).long(), i2.reshape(-1).long()

    dtype = property(lambda self: self.ktt.dtype)
    device = property(lambda self: self.ktt.device)

    def _size(self):
        return torch.Size([self.i1.numel(), self.i2.numel()])

    def to_dense(self):
        return self.ktt[self.i1][:, self.i2]

    def _matmul(self, rhs):
        return self.to_dense() @ rhs

    def _transpose_nonbatch(self):
        return IndexedTaskCovar(self.ktt.mT, self.i2, self.i1)

    def diagonal(self, offset=0, dim1=-2, dim2=-1):
        return self.ktt[self.i1, self.i2]

    def __getitem__(self, index):
        from .operators import _strip_ellipsis

        index = _strip_ellipsis(index)
        r, c = index if isinstance(index, tuple) else (index, slice(None))
        return IndexedTaskCovar(self.ktt, self.i1[r], self.i2[c])

    def detach(self):
        return IndexedTaskCovar(self.ktt.detach(), self.i1, self.i2)

    def mul(self, other):
        if isinstance(other, FusedKernelLinearOperator):
            return HadamardFusedLinearOperator(other, self)
        return super().mul(other)

    __mul__ = mul


def hadamard_matvec(p1, p2, os_, ktt, i1, i2, vt):
    """((os * k(x1, x2)) o K_TT[i1, i2]) @ V for V = vt [t, >= m] probe-major; returns [t, ld_n]."""
    T, t = ktt.shape[-1], vt.shape[0]
    n, m = p1.n, p2.n
    ldm = B.round_up(m, 4)
    mask2 = torch.nn.functional.one_hot(i2, T).t().to(vt.dtype)                    # [T, m]
    w = torch.zeros(t, T, ldm, device=vt.device, dtype=vt.dtype)
    w[:, :, :m] = vt[:, None, :m] * mask2[None]
    q = B.kv(p1, p2, w.reshape(t * T, ldm), scale=os_)                              # [(t T), ld_n]
    bsel = ktt.to(vt.dtype)[i1].t()                                                 # [T, n]: K_TT[task_i, tau']
    out = torch.zeros(t, q.shape[1], device=vt.device, dtype=vt.dtype)
    out[:, :n] = (q[:, :n].reshape(t, T, n) * bsel[None]).sum(1)
    return out


class HadamardFusedLinearOperator(LinearOperator):
    """(outputscale * k(x1, x2)) o K_TT[i1, i2], matrix-free."""

    def __init__(self, kx: FusedKernelLinearOperator, tasks: IndexedTaskCovar):
        if kx.shape != tasks.shape:
            raise RuntimeError(f"data covariance {tuple(kx.shape)} and task covariance {tuple(tasks.shape)} sizes do not match")
        self.kx, self.tasks = kx, tasks

    dtype = property(lambda self: self.kx.dtype)
    device = property(lambda self: self.kx.device)

    @property
    def requires_grad(self):
        return self.kx.requires_grad or self.tasks.ktt.requires_grad

    def _size(self):
        return self.kx._size()

    def _matmul(self, rhs):
        p1, p2 = self.kx.prepared()
        out_t = hadamard_matvec(p1, p2, self.kx._os(), self.tasks.ktt.detach(), self.tasks.i1, self.tasks.i2, B.to_probe_major(rhs.detach(), p1.dtype))
        return B.from_probe_major(out_t, self.shape[0]).to(rhs.dtype)

    def _transpose_nonbatch(self):
        return HadamardFusedLinearOperator(self.kx._transpose_nonbatch(), self.tasks._transpose_nonbatch())

    def _mul_constant(self, c):
        return HadamardFusedLinearOperator(self.kx._mul_constant(c), self.tasks)

    def diagonal(self, offset=0, dim1=-2, dim2=-1):
        return self.kx.diagonal() * self.tasks.diagonal()

    def to_dense(self):
        return self.kx.to_dense() * self.tasks.to_dense().to(self.dtype)

    def __getitem__(self, index):
        a, b = self.kx[index], self.tasks[index]
        if isinstance(a, FusedKernelLinearOperator):
            return HadamardFusedLinearOperator(a, b)
        return a * b.to_dense()

    def detach(self):
        return HadamardFusedLinearOperator(self.kx.detach(), self.tasks.detach())

    def __add__(self, other):
        if isinstance(other, DiagLinearOperator) and self.is_square and not other.batch_shape:
            noise, vec = split_diag(other, self.device, self.dtype)
            return HadamardFusedAddedDiagLinearOperator(self, noise, noise_vec=vec)
        return super().__add__(other)


class HadamardFusedAddedDiagLinearOperator(LinearOperator):
    """(k o K_TT) + noise I: the operator the MLL and the prediction caches of a Hadamard multitask GP solve with."""

    def __init__(self, had: HadamardFusedLinearOperator, noise: torch.Tensor, noise_vec=None, bbmm_opts=None):
        self.had = had
        self.noise = noise.reshape(-1)[:1]
        self.noise_vec = noise_vec
        self.bbmm_opts = {} if bbmm_opts is None else bbmm_opts

    dtype = property(lambda self: self.had.dtype)
    device = property(lambda self: self.had.device)

    @property
    def requires_grad(self):
        return self.had.requires_grad or self.noise.requires_grad

    def _size(self):
        return self.had._size()

    def _diag_total(self):
        d = self.noise.reshape(()).expand(self.shape[-1])
        return d if self.noise_vec is None else d + self.noise_vec

    def _matmul(self, rhs):
        return self.had._matmul(rhs) + self._diag_total().detach().unsqueeze(-1) * rhs

    def _transpose_nonbatch(self):
        return self

    def diagonal(self, offset=0, dim1=-2, dim2=-1):
        return self.had.diagonal() + self._diag_total()

    def to_dense(self):
        return self.had.to_dense() + torch.diag(self._diag_total().to(self.dtype))

    def detach(self):
        return HadamardFusedAddedDiagLinearOperator(self.had.detach(), self.noise.detach(), self.noise_vec, self.bbmm_opts)

    def __add__(self, other):
        if isinstance(other, DiagLinearOperator) and not other.batch_shape:
            noise, vec = split_diag(other, self.device, self.dtype)
            nv = self.noise_vec if vec is None else (vec if self.noise_vec is None else self.noise_vec + vec)
            return HadamardFusedAddedDiagLinearOperator(self.had, self.noise + noise, nv, self.bbmm_opts)
        return super().__add__(other)

    def _use_cholesky(self, flag):
        return flag.off() or self.shape[-1] <= settings.max_cholesky_size.value()

    def _dvec(self, wd):
        n = self.shape[-1]
        dv = torch.zeros(B.round_up(n, 4), device=self.device, dtype=wd)
        dv[:n] = self._diag_total().detach().to(wd)
        return dv

    def _partials(self):
        p1, _ = self.had.kx.prepared()
        os_, ktt, ti = self.had.kx._os(), self.had.tasks.ktt.detach(), self.had.tasks.i1

        def partials(dt):
            out = hadamard_matvec(p1, p1, os_, ktt, ti, ti, dt)
            return out, 1, out.stride(0)

        return partials, p1.dtype

    def inv_quad_logdet(self, inv_quad_rhs=None, logdet=False, reduce_inv_quad=True):
        n = self.shape[-1]
        if inv_quad_rhs is None:
            inv_quad_rhs = torch.zeros(n, 0, device=self.device, dtype=self.dtype)
        rhs = inv_quad_rhs.unsqueeze(-1) if inv_quad_rhs.dim() == 1 else inv_quad_rhs
        if self._use_cholesky(settings.fast_computations.log_prob):
            Lc = psd_safe_cholesky(self.to_dense().to(torch.float64), model_dtype=self.dtype)   # members' dense kernels are autograd-visible
            sol = torch.cholesky_solve(rhs.to(torch.float64), Lc)
            iq = (sol * rhs.to(torch.float64)).sum(-2).to(rhs.dtype)
            ld = (2.0 * Lc.diagonal().log().sum()).to(rhs.dtype)
        else:
            drop = rhs.shape[-1] == 0
            if drop:
                rhs = torch.zeros(n, 1, device=self.device, dtype=self.dtype)
            kx = self.had.kx
            wd = B.work_dtype(kx.x1)
            nvec = None if self.noise_vec is None else self.noise_vec.detach().to(wd)
            iq, ld = HadamardInvQuadLogdetFn.apply(kx.x1, kx.lengthscale, kx.outputscale, self.had.tasks.ktt, self.noise, rhs, kx.spec,
                                                   self.had.tasks.i1, nvec, self.bbmm_opts, kx.spec.param)
            if drop:
                iq = iq[:0]
        if reduce_inv_quad:
            iq = iq.sum(-1)
        return iq, (ld if logdet else None)

    # ---- float64 product on the prepared points of the float32 path (mixed-precision corrections: settings.rhs_refinement; round 6) ----
    def float64_product_available(self) -> bool:
        p1, _ = self.had.kx.prepared()
        return bool(p1.fused and p1.dp <= B.FUSED_F64_MAX_DP)

    def _matvec64(self):
        """a64 [c, ld] (probe-major, float64) -> ((theta K_XX) o K_TT[i, i] + D) a in float64: ``hadamard_matvec`` on the prepared points widened to
        float64 (the fused float64 product with T x columns, ``csrc/kv_f64.hpp``), groups of at most 80 / T columns."""
        p1, _ = self.had.kx.prepared()
        x64 = B.PreparedPoints(p1.xp.to(torch.float64), p1.n, p1.d, p1.dp, p1.kind, p1.param)
        os_ = self.had.kx._os()
        os64 = None if os_ is None else os_.detach().to(torch.float64)
        ktt, ti = self.had.tasks.ktt.detach().to(torch.float64), self.had.tasks.i1
        dv = self._dvec(torch.float64)
        grp = max(1, 80 // ktt.shape[-1])

        def mv(a64):
            out = torch.empty_like(a64)
            for c0 in range(0, a64.shape[0], grp):
                blk = a64[c0 : c0 + grp].contiguous()
                out[c0 : c0 + grp] = hadamard_matvec(x64, x64, os64, ktt, ti, ti, blk)[:, : blk.shape[1]] + dv.unsqueeze(0)[:, : blk.shape[1]] * blk
            return out

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
            dv = self._dvec(wd)
            if not hasattr(self, "_precond_cache"):
                p1, _ = self.had.kx.prepared()
                self._precond_cache = hadamard_preconditioner(p1, self.had.kx._os(), self.had.tasks.ktt, self.had.tasks.i1, dv, self.shape[-1])

            def cg32(rt):
                return linear_cg(None, None, None, rt, n_tridiag=0, tolerance=settings.cg_tolerance.value(),
                                 kv_partials=partials, dvec=dv, nvec=self.shape[-1], preconditioner=self._precond_cache)

            rhs_t = B.to_probe_major(r.detach(), wd)
            sol_t, _ = cg32(rhs_t)
            if settings.rhs_refinement.on() and sol_t.dtype == torch.float32 and self.float64_product_available():
                # mixed-precision refinement on the Hadamard operator too (round 6): float64 residual through hadamard_matvec on the widened points
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
        method = check_root_method(method, inverse=True)
        if method in ("cholesky", "symeig") or (method is None and self._use_cholesky(settings.fast_computations.covar_root_decomposition)):
            return super().root_inv_decomposition(method=method)      # dense factorisations of a small operator (base class)
        n = self.shape[-1]
        partials, wd = self._partials()
        dv = self._dvec(wd)

        def mv(q_row):
            out, _, _ = partials(q_row)
            return out + dv.unsqueeze(0) * q_row

        init_t, test_t = lanczos_vectors(initial_vectors, test_vectors, n, wd)
        rt = root_inv_decomposition(None, None, None, matvec=mv, nvec=n, device=self.device, generator=self.bbmm_opts.get("generator"),
                                    init_vec_t=init_t, test_vec_t=test_t, dtype=wd)
        return RootLinearOperator(B.from_probe_major(rt, n).to(self.dtype))


def hadamard_preconditioner(xp, os_, ktt, ti, diag_total, n, rank=None, tol=None, min_size=None):
    """Pivoted-Cholesky preconditioner of (theta k(x, x)) o K_TT[ti, ti] + diag: row p is the data-kernel row of point p
    (``gpamd_kernel_rows_f32``) times K_TT[task_p, task_.] (the reference preconditions the product operator + diagonal like any other
    ``AddedDiagLinearOperator``: ``kernels/index_kernel.py:101-112`` + ``settings.py:6-31``)."""
    if not xp.fused:
        return None
    wd = xp.dtype
    ktt_d = ktt.detach().to(wd)

    def row_fn(p):
        return B.kernel_rows(xp, p, xp, os_).reshape(-1) * ktt_d[ti[p]].reshape(-1)[ti]

    kdiag = B.kernel_diag(xp, xp, os_) * ktt_d[ti, ti]
    d = diag_total.detach()[:n].to(wd)
    if bool((d == d[0]).all()):
        return build_preconditioner_rows(row_fn, kdiag, d[:1], False, rank, tol, min_size)
    return build_preconditioner_rows(row_fn, kdiag, d, True, rank, tol, min_size)


class HadamardInvQuadLogdetFn(torch.autograd.Function):
    """(inv_quad[c], logdet) of (theta k(x, x)) o K_TT[ti, ti] + noise I (+ diag(noise_vec)) by preconditioned mBCG + SLQ
    (``bbmm.inv_quad_logdet_forward`` on the masked-column product; probe columns shardable over ``opts["group"]``)."""

    @staticmethod
    def forward(ctx, x, lengthscale, outputscale, ktt, noise, rhs, spec: KernelSpec, ti, noise_vec, opts, kparam=None):
        n, T = x.shape[-2], ktt.shape[-1]
        dev = x.device
        xp = _prep(spec, x, lengthscale)
        wd = xp.dtype
        os_ = None if outputscale is None else outputscale.detach().reshape(-1)[:1].to(wd).contiguous()
        ktt_d = ktt.detach().to(wd)
        opts_in = opts
        opts = structured_opts(opts, dev)
        ld = B.round_up(n, 4)
        dv = torch.zeros(ld, device=dev, dtype=wd)
        dv[:n] = noise.detach().reshape(()).to(wd)
        if noise_vec is not None:
            dv[:n] += noise_vec

        def partials(dt):
            out = hadamard_matvec(xp, xp, os_, ktt_d, ti, ti, dt)
            return out, 1, out.stride(0)

        pre = opts.get("precond", "auto")
        if pre == "auto":
            pre = hadamard_preconditioner(xp, os_, ktt_d, ti, dv, n)
        res = inv_quad_logdet_forward(
            None, None, None, B.to_probe_major(rhs, wd), num_probes=opts.get("num_probes"), precond=pre, probes=opts.get("probes"),
            generator=opts.get("generator"), tolerance=opts.get("tolerance"), max_iter=opts.get("max_iter"), group=opts.get("group"),
            t_total=opts.get("t_total"), dvec=dv, kv_partials=partials, nvec=n,
        )
        ctx.xp, ctx.n, ctx.T, ctx.ti, ctx.res = xp, n, T, ti, res
        ctx.kparam = kparam
        ctx.group = opts.get("group")
        ctx.t_total = opts.get("t_total") or res.zt.shape[0]
        ctx.has_os = outputscale is not None
        ctx.save_for_backward(lengthscale, outputscale if outputscale is not None else torch.empty(0), ktt, noise, rhs)
        opts_in["_last_info"] = res.info
        return res.inv_quad.to(rhs.dtype), res.logdet.to(rhs.dtype)

    @staticmethod
    def backward(ctx, g_iq, g_ld):
        lengthscale, outputscale, ktt, noise, rhs = ctx.saved_tensors
        outputscale = outputscale if ctx.has_os else None
        xp, n, T, ti, res = ctx.xp, ctx.n, ctx.T, ctx.ti, ctx.res
        wd = xp.dtype
        left, right, s_y = backward_vectors(res, g_iq, g_ld, ctx.t_total)
        c = s_y.shape[0]
        tc = left.shape[0]
        left, right = left[:, :n], right[:, :n]
        ktt_d = ktt.detach().to(wd)
        mask = torch.nn.functional.one_hot(ti, T).t().to(wd)            # [T, n]
        ld = B.round_up(n, 4)

        def pad(v3):                                                    # [tc, T, n] -> probe-major [(tc T), ld]
            out = torch.zeros(tc * T, ld, device=v3.device, dtype=wd)
            out[:, :n] = v3.reshape(tc * T, n)
            return out

        l_m = left[:, None, :] * mask[None]                             # L_c o 1[task = tau]
        r_b = right[:, None, :] * ktt_d[:, ti][None]                    # R_c[j] * K_TT[tau, task_j]
        kp = ctx.kparam if (ctx.kparam is not None and ctx.needs_input_grad[10]) else None
        out = hyper_grads(xp, xp, lengthscale, outputscale, pad(l_m), pad(r_b), kparam=kp)
        d_ls, d_os = out[:2]
        d_par = out[-1] if kp is not None else None
        # d/dK_TT[tau, tau'] = sum_c (L_c o 1[tau])^T (theta K) (R_c o 1[tau'])
        os_ = None if outputscale is None else outputscale.detach().reshape(-1)[:1].to(wd).contiguous()
        mq = B.kv(xp, xp, pad(right[:, None, :] * mask[None]), scale=os_)[:, :n].reshape(tc, T, n)
        d_ktt = torch.einsum("cti,csi->ts", l_m, mq).to(ktt.dtype)
        d_noise = (left * right).sum().reshape(noise.shape).to(noise.dtype)
        allreduce_grads_([d_ls, d_os, d_ktt, d_noise, d_par], ctx.group)
        d_rhs = (2.0 * B.from_probe_major(s_y, n) * g_iq.to(wd).reshape(1, c)).to(rhs.dtype) if ctx.needs_input_grad[5] else None
        return None, d_ls, d_os, d_ktt, d_noise, d_rhs, None, None, None, None, d_par
