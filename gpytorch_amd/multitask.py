"""Multitask exact GPs on the fused path (BASELINE config 5).

Mirrors ``gpytorch/kernels/index_kernel.py:91-112`` (K_TT = B B^T + diag(v)),
``gpytorch/kernels/multitask_kernel.py:46-61`` (K = K_XX (x) K_TT as a Kronecker operator),
``gpytorch/means/multitask_mean.py:41-45``, ``gpytorch/distributions/multitask_multivariate_normal.py:34-71``
(interleaved layout: row = i*T + tau) and ``gpytorch/likelihoods/multitask_gaussian_likelihood.py:76-154``
(noise I_n (x) (D_T + s2 I_T)).

The reference solves K_XX (x) K_TT + noise through per-factor eigendecompositions of K_XX -- O(n^3), infeasible
at n = 2e5.  Here, as BASELINE config 5 prescribes, the system is solved by batched CG with the Kronecker MVM
    vec^-1(V) -> K_XX . V_mat . K_TT^T
whose n^2 part is ONE fused K*V launch with T*t columns (columns = (task, probe) pairs); the T x T factor and the
(de)interleaving are O(n T t) glue.
"""
from __future__ import annotations

import torch

from . import backend as B
from . import settings
from .bbmm import allreduce_grads_, backward_vectors, build_preconditioner_rows, inv_quad_logdet_forward, structured_opts
from .distributions import MultivariateNormal
from .functions import KernelSpec, _prep, hyper_grads
from .kernels import Kernel
from .lanczos import root_inv_decomposition
from .likelihoods import _GaussianLikelihoodBase
from .linear_cg import linear_cg
from .means import Mean
from .module import GreaterThan, Module, Positive
from .operators import DiagLinearOperator, FusedKernelLinearOperator, LinearOperator, RootLinearOperator, check_root_method, lanczos_vectors, psd_safe_cholesky


# ------------------------------------------------------------------------------------------------ layout helpers
def _deinterleave(vt: torch.Tensor, n: int, T: int) -> torch.Tensor:
    """[t, >= n*T] interleaved (i*T + tau)  ->  probe-major [(t*T), ld_n] with row (c, tau)."""
    t = vt.shape[0]
    out = torch.zeros(t * T, B.round_up(n, 4), device=vt.device, dtype=vt.dtype)
    out[:, :n] = vt[:, : n * T].reshape(t, n, T).permute(0, 2, 1).reshape(t * T, n)
    return out


def _interleave(q: torch.Tensor, t: int, n: int, T: int) -> torch.Tensor:
    """[(t*T), >= n] rows (c, tau)  ->  [t, ld_{nT}] interleaved."""
    out = torch.zeros(t, B.round_up(n * T, 4), device=q.device, dtype=q.dtype)
    out[:, : n * T] = q[:, :n].reshape(t, T, n).permute(0, 2, 1).reshape(t, n * T)
    return out


def kron_matvec(p1: B.PreparedPoints, p2: B.PreparedPoints, ktt: torch.Tensor, vt: torch.Tensor, scale=None) -> torch.Tensor:
    """(scale * K(x1,x2) (x) K_TT) @ V for V = vt [t, >= n2*T] interleaved; returns [t, ld_{n1 T}] interleaved."""
    T = ktt.shape[-1]
    t = vt.shape[0]
    w = _deinterleave(vt.to(p1.dtype), p2.n, T)
    q = B.kv(p1, p2, w, scale=scale)  # [(t*T), ld_n1]: K_XX @ V_mat, one fused launch with T*t columns
    q3 = torch.einsum("ab,cbn->can", ktt.to(q.dtype), q[:, : p1.n].reshape(t, T, p1.n))
    return _interleave(q3.reshape(t * T, p1.n), t, p1.n, T)


# ------------------------------------------------------------------------------------------------ operators
class KroneckerFusedLinearOperator(LinearOperator):
    """K_XX (x) K_TT with K_XX a :class:`FusedKernelLinearOperator` and K_TT a dense T x T tensor."""

    def __init__(self, kx: FusedKernelLinearOperator, ktt: torch.Tensor):
        self.kx, self.ktt = kx, ktt
        self.T = ktt.shape[-1]

    dtype = property(lambda self: self.kx.dtype)
    device = property(lambda self: self.kx.device)

    def _size(self):
        s = self.kx.shape
        return torch.Size([s[0] * self.T, s[1] * self.T])

    def _matmul(self, rhs):
        p1, p2 = self.kx.prepared()
        out_t = kron_matvec(p1, p2, self.ktt.detach(), B.to_probe_major(rhs, p1.dtype), self.kx._os())
        return B.from_probe_major(out_t, self.shape[0]).to(rhs.dtype)

    def _transpose_nonbatch(self):
        return KroneckerFusedLinearOperator(self.kx._transpose_nonbatch(), self.ktt.mT)

    def diagonal(self, offset=0, dim1=-2, dim2=-1):
        return (self.kx.diagonal().unsqueeze(-1) * self.ktt.diagonal().unsqueeze(0)).reshape(-1)

    def to_dense(self):
        return torch.kron(self.kx.to_dense(), self.ktt.to(self.dtype))

    def __getitem__(self, index):
        if isinstance(index, tuple) and len(index) == 3 and index[0] is Ellipsis:
            index = index[1:]
        r, c = index if isinstance(index, tuple) else (index, slice(None))
        T = self.T

        def data_slice(sl, n):
            a, b, step = sl.indices(n * T)
            if step != 1 or a % T or b % T:
                raise NotImplementedError("Kronecker operator slices must be aligned to whole data points")
            return slice(a // T, b // T)

        if isinstance(r, slice) and isinstance(c, slice):
            return KroneckerFusedLinearOperator(self.kx[data_slice(r, self.kx.shape[0]), data_slice(c, self.kx.shape[1])], self.ktt)
        return self.to_dense()[index]

    def __add__(self, other):
        if isinstance(other, TaskNoiseDiagLinearOperator) and self.is_square:
            return KroneckerFusedAddedDiagLinearOperator(self, other.task_noise)
        return super().__add__(other)

    def detach(self):
        return KroneckerFusedLinearOperator(self.kx.detach(), self.ktt.detach())


class TaskNoiseDiagLinearOperator(DiagLinearOperator):
    """I_n (x) diag(task_noise): the noise covariance of MultitaskGaussianLikelihood (rank 0)."""

    def __init__(self, task_noise: torch.Tensor, n: int):
        self.task_noise, self.n = task_noise, n

    @property
    def _diag(self):
        return self.task_noise.repeat(self.n)


class KroneckerFusedAddedDiagLinearOperator(LinearOperator):
    """K_XX (x) K_TT + I_n (x) diag(task_noise): solved by mBCG over the Kronecker MVM."""

    def __init__(self, kron: KroneckerFusedLinearOperator, task_noise: torch.Tensor, bbmm_opts=None):
        self.kron, self.task_noise = kron, task_noise
        self.bbmm_opts = {} if bbmm_opts is None else bbmm_opts
        self._cache = {}

    dtype = property(lambda self: self.kron.dtype)
    device = property(lambda self: self.kron.device)

    def _size(self):
        return self.kron._size()

    def _transpose_nonbatch(self):
        return self

    def _dvec(self):
        n, T = self.kron.kx.shape[0], self.kron.T
        wd = B.work_dtype(self.kron.kx.x1)
        dv = torch.zeros(B.round_up(n * T, 4), device=self.device, dtype=wd)
        dv[: n * T] = self.task_noise.detach().to(wd).repeat(n)
        return dv

    def _matmul(self, rhs):
        return self.kron._matmul(rhs) + self.task_noise.detach().repeat(self.kron.kx.shape[0]).unsqueeze(-1) * rhs

    def diagonal(self, offset=0, dim1=-2, dim2=-1):
        return self.kron.diagonal() + self.task_noise.repeat(self.kron.kx.shape[0])

    def to_dense(self):
        return self.kron.to_dense() + torch.diag(self.task_noise.repeat(self.kron.kx.shape[0]).to(self.dtype))

    def detach(self):
        return KroneckerFusedAddedDiagLinearOperator(self.kron.detach(), self.task_noise.detach(), self.bbmm_opts)

    def __add__(self, other):
        if isinstance(other, TaskNoiseDiagLinearOperator):
            return KroneckerFusedAddedDiagLinearOperator(self.kron, self.task_noise + other.task_noise, self.bbmm_opts)
        return super().__add__(other)

    def _use_cholesky(self, flag):
        return flag.off() or self.shape[-1] <= settings.max_cholesky_size.value()

    def _precond(self):
        """The pivoted-Cholesky preconditioner of this operator (``kron_preconditioner``), built once per operator."""
        if "precond" not in self._cache:
            p1, _ = self.kron.kx.prepared()
            self._cache["precond"] = kron_preconditioner(p1, self.kron.kx._os(), self.kron.ktt, self.task_noise) if p1.fused else None
        return self._cache["precond"]

    def _cg(self, rhs_t, n_tridiag=0, tolerance=None):
        p1, _ = self.kron.kx.prepared()
        ktt, os_ = self.kron.ktt.detach(), self.kron.kx._os()
        N = self.shape[-1]

        def partials(dt):
            out = kron_matvec(p1, p1, ktt, dt, os_)
            return out, 1, out.stride(0)

        return linear_cg(None, None, None, rhs_t, n_tridiag=n_tridiag, tolerance=tolerance, kv_partials=partials,
                         dvec=self._dvec(), nvec=N, group=self.bbmm_opts.get("group"), preconditioner=self._precond())

    # ---- float64 product on the prepared points of the float32 path (mixed-precision corrections: settings.rhs_refinement) ----
    def float64_product_available(self) -> bool:
        p1, _ = self.kron.kx.prepared()
        return bool(p1.fused and p1.dp <= B.FUSED_F64_MAX_DP)

    def _matvec64(self):
        """a64 [c, ld_{nT}] (probe-major, interleaved, float64) -> (K_XX (x) K_TT + I (x) diag(task noise)) a in float64: the Kronecker MVM of
        ``kron_matvec`` on the prepared points widened to float64 (one fused float64 launch with T x columns, ``csrc/kv_f64.hpp``), groups of 20 columns."""
        p1, _ = self.kron.kx.prepared()
        x64 = B.PreparedPoints(p1.xp.to(torch.float64), p1.n, p1.d, p1.dp, p1.kind, p1.param)
        ktt = self.kron.ktt.detach().to(torch.float64)
        os_ = self.kron.kx._os()
        os64 = None if os_ is None else os_.to(torch.float64)
        dv = self._dvec().to(torch.float64)
        grp = max(1, 80 // self.kron.T)

        def mv(a64):
            out = torch.empty_like(a64)
            for c0 in range(0, a64.shape[0], grp):
                blk = a64[c0 : c0 + grp].contiguous()
                out[c0 : c0 + grp] = kron_matvec(x64, x64, ktt, blk, os64)[:, : blk.shape[1]] + dv.unsqueeze(0)[:, : blk.shape[1]] * blk
            return out

        return mv

    def matmul_float64(self, rhs: torch.Tensor) -> torch.Tensor:
        """K_hat @ rhs ([nT, c] -> [nT, c]) in float64 (see ``FusedKernelAddedDiagLinearOperator.matmul_float64``)."""
        if not self.float64_product_available():
            return None
        return B.from_probe_major(self._matvec64()(B.to_probe_major(rhs.detach(), torch.float64)), self.shape[-1])

    def solve(self, rhs, lhs=None):
        squeeze = rhs.dim() == 1
        r = rhs.unsqueeze(-1) if squeeze else rhs
        if self._use_cholesky(settings.fast_computations.solves):
            sol = torch.cholesky_solve(r.detach().double(), psd_safe_cholesky(self.to_dense().detach().double(), model_dtype=self.dtype)).to(rhs.dtype)
        else:
            rhs_t = B.to_probe_major(r.detach(), B.work_dtype(self.kron.kx.x1))
            sol_t, info = self._cg(rhs_t, tolerance=settings.cg_tolerance.value())
            self._cache["last_cg_info"] = info
            if settings.rhs_refinement.on() and sol_t.dtype == torch.float32 and self.bbmm_opts.get("group") is None and self.float64_product_available():
                # mixed-precision refinement of the float32 solves (round 6: the structured operators too): float64 residual through the Kronecker
                # MVM on the same prepared points, one more float32 solve of it (bbmm.refine_with_)
                from .bbmm import refine_with_

                def solve32(res):
                    d, inf = self._cg(res, tolerance=settings.cg_tolerance.value())
                    return d, inf.iterations

                refine_with_(rhs_t.to(torch.float64), sol_t, self._matvec64(), solve32, settings.rhs_refinement.steps)
            sol = B.from_probe_major(sol_t, self.shape[-1]).to(rhs.dtype)
        if lhs is not None:
            sol = lhs @ sol
        return sol.squeeze(-1) if squeeze else sol

    def inv_quad_logdet(self, inv_quad_rhs=None, logdet=False, reduce_inv_quad=True):
        kx = self.kron.kx
        N = self.shape[-1]
        rhs = inv_quad_rhs.unsqueeze(-1) if inv_quad_rhs.dim() == 1 else inv_quad_rhs
        if self._use_cholesky(settings.fast_computations.log_prob):
            K = self.to_dense_differentiable()
            Lc = psd_safe_cholesky(K.double(), model_dtype=K.dtype)
            sol = torch.cholesky_solve(rhs.double(), Lc)
            iq = (sol * rhs.double()).sum(-2).to(rhs.dtype)
            ld = (2.0 * Lc.diagonal().log().sum()).to(rhs.dtype)
        else:
            iq, ld = KroneckerInvQuadLogdetFn.apply(kx.x1, kx.lengthscale, kx.outputscale, self.kron.ktt, self.task_noise, rhs,
                                                    kx.spec, self.bbmm_opts, kx.spec.param)
        if reduce_inv_quad:
            iq = iq.sum(-1)
        _ = N
        return iq, (ld if logdet else None)

    def to_dense_differentiable(self):
        """Small-n Cholesky branch: dense K_hat built with autograd-visible torch ops on the device."""
        kx = self.kron.kx
        z1 = (kx.x1 - kx.spec.shift) / kx.lengthscale
        d2 = (z1.unsqueeze(-2) - z1.unsqueeze(-3)).pow(2).sum(-1)
        if kx.spec.kind == "rbf":
            kmat = torch.exp(-0.5 * d2)
        elif kx.spec.kind == "rq":
            alpha = kx.spec.param.reshape(())
            kmat = (1 + d2 / (2 * alpha)).pow(-alpha)
        else:
            nu = {"matern12": 0.5, "matern32": 1.5, "matern52": 2.5}[kx.spec.kind]
            r = (d2 + 1e-20).sqrt() * (2 * nu) ** 0.5
            e = torch.exp(-r)
            kmat = e if nu == 0.5 else ((1 + r) * e if nu == 1.5 else (1 + r + r * r / 3) * e)
        if kx.outputscale is not None:
            kmat = kmat * kx.outputscale.reshape(())
        return torch.kron(kmat, self.kron.ktt) + torch.diag(self.task_noise.repeat(kx.shape[0]))

    def root_inv_decomposition(self, initial_vectors=None, test_vectors=None, method=None):
        method = check_root_method(method, inverse=True)
        if method in ("cholesky", "symeig") or (method is None and self._use_cholesky(settings.fast_computations.covar_root_decomposition)):
            return super().root_inv_decomposition(method=method)      # dense factorisations of a small operator (base class)
        p1, _ = self.kron.kx.prepared()
        ktt, os_, dv = self.kron.ktt.detach(), self.kron.kx._os(), self._dvec()
        N = self.shape[-1]

        def mv(q_row):
            return kron_matvec(p1, p1, ktt, q_row, os_) + dv.unsqueeze(0) * q_row

        init_t, test_t = lanczos_vectors(initial_vectors, test_vectors, N, p1.dtype)
        rt = root_inv_decomposition(None, None, None, matvec=mv, nvec=N, device=self.device, init_vec_t=init_t, test_vec_t=test_t, dtype=p1.dtype)
        return RootLinearOperator(B.from_probe_major(rt, N).to(self.dtype))


def kron_preconditioner(xp, os_, ktt, task_noise, rank=None, tol=None, min_size=None):
    """Pivoted-Cholesky preconditioner of K_XX (x) K_TT + I (x) diag(task_noise) (the reference preconditions this operator like any
    other ``AddedDiagLinearOperator``: ``kernels/multitask_kernel.py:46-54`` + ``settings.py:6-31``).  Row (i, tau) of the Kronecker
    product is theta k(x_i, .) (x) K_TT[tau, :] -- one ``gpamd_kernel_rows_f32`` row per step; the per-task noise makes the diagonal
    non-constant unless all task noises agree (``Preconditioner.dinv_sqrt``)."""
    n, T = xp.n, ktt.shape[-1]
    wd = xp.dtype
    ktt_d = ktt.detach().to(wd)
    theta = 1.0 if os_ is None else os_.reshape(())

    def row_fn(p):
        i, tau = torch.div(p, T, rounding_mode="floor"), p % T
        kr = B.kernel_rows(xp, i, xp, os_).reshape(n, 1)             # theta k(x_i, x_j), j = 0..n-1
        return (kr * ktt_d[tau].reshape(1, T)).reshape(-1)           # interleaved (j T + tau')

    kdiag = (B.kernel_diag(xp, xp, os_).reshape(n, 1) * ktt_d.diagonal().reshape(1, T)).reshape(-1)
    _ = theta
    tn = task_noise.detach().to(wd).reshape(-1)
    if bool((tn == tn[0]).all()):
        return build_preconditioner_rows(row_fn, kdiag, tn[:1], False, rank, tol, min_size)
    return build_preconditioner_rows(row_fn, kdiag, tn.repeat(n), True, rank, tol, min_size)


class KroneckerInvQuadLogdetFn(torch.autograd.Function):
    """inv_quad / log-det of K_XX (x) K_TT + I (x) diag(task_noise) by preconditioned mBCG + SLQ (``bbmm.inv_quad_logdet_forward`` on
    the Kronecker product -- probe columns sharded over ``opts["group"]`` exactly like the single-kernel operator: BASELINE C5,
    "batched CG over task blocks, 4 x MI355X"), with the A.6 backward specialised to the Kronecker structure (one fused
    bilinear-derivative launch + one fused K*V launch; gradients summed over the probe group in one packed all-reduce)."""

    @staticmethod
    def forward(ctx, x, lengthscale, outputscale, ktt, task_noise, rhs, spec: KernelSpec, opts: dict, kparam=None):
        n, T = x.shape[-2], ktt.shape[-1]
        N = n * T
        dev = x.device
        xp = _prep(spec, x, lengthscale)
        wd = xp.dtype
        os_ = None if outputscale is None else outputscale.detach().reshape(-1)[:1].to(wd).contiguous()
        ktt_d = ktt.detach().to(wd)
        opts_in = opts
        opts = structured_opts(opts, dev)
        ld = B.round_up(N, 4)
        dv = torch.zeros(ld, device=dev, dtype=wd)
        dv[:N] = task_noise.detach().to(wd).repeat(n)

        def partials(dt):
            out = kron_matvec(xp, xp, ktt_d, dt, os_)
            return out, 1, out.stride(0)

        pre = opts.get("precond", "auto")
        if pre == "auto":
            pre = kron_preconditioner(xp, os_, ktt_d, task_noise)
        res = inv_quad_logdet_forward(
            None, None, None, B.to_probe_major(rhs, wd), num_probes=opts.get("num_probes"), precond=pre, probes=opts.get("probes"),
            generator=opts.get("generator"), tolerance=opts.get("tolerance"), max_iter=opts.get("max_iter"), group=opts.get("group"),
            t_total=opts.get("t_total"), dvec=dv, kv_partials=partials, nvec=N,
        )
        ctx.xp, ctx.n, ctx.T, ctx.res = xp, n, T, res
        ctx.kparam = kparam
        ctx.group = opts.get("group")
        ctx.t_total = opts.get("t_total") or res.zt.shape[0]
        ctx.has_os = outputscale is not None
        ctx.save_for_backward(lengthscale, outputscale if outputscale is not None else torch.empty(0), ktt, task_noise, rhs)
        opts_in["_last_info"] = res.info
        return res.inv_quad.to(rhs.dtype), res.logdet.to(rhs.dtype)

    @staticmethod
    def backward(ctx, g_iq, g_ld):
        lengthscale, outputscale, ktt, task_noise, rhs = ctx.saved_tensors
        outputscale = outputscale if ctx.has_os else None
        xp, n, T, res = ctx.xp, ctx.n, ctx.T, ctx.res
        N = n * T
        wd = xp.dtype
        left, right, s_y = backward_vectors(res, g_iq, g_ld, ctx.t_total)
        c = s_y.shape[0]
        tc = left.shape[0]
        l3 = left[:, :N].reshape(tc, n, T).permute(0, 2, 1).contiguous()   # [tc, T, n]
        r3 = right[:, :N].reshape(tc, n, T).permute(0, 2, 1).contiguous()
        ktt32 = ktt.detach().to(wd)
        r3k = torch.einsum("ab,cbn->can", ktt32, r3)
        ld_n = B.round_up(n, 4)

        def pad(v3):
            out = torch.zeros(tc * T, ld_n, device=v3.device, dtype=wd)
            out[:, :n] = v3.reshape(tc * T, n)
            return out

        lp = pad(l3)
        kp = ctx.kparam if (ctx.kparam is not None and ctx.needs_input_grad[8]) else None
        out = hyper_grads(xp, xp, lengthscale, outputscale, lp, pad(r3k), kparam=kp)
        d_ls, d_os = out[:2]
        d_par = out[-1] if kp is not None else None
        os_ = None if outputscale is None else outputscale.detach().reshape(-1)[:1].to(wd).contiguous()
        m3 = B.kv(xp, xp, pad(r3), scale=os_)[:, :n].reshape(tc, T, n)
        d_ktt = torch.einsum("ctn,csn->ts", l3, m3).to(ktt.dtype)
        d_noise = (l3 * r3).sum(dim=(0, 2)).to(task_noise.dtype)
        allreduce_grads_([d_ls, d_os, d_ktt, d_noise, d_par], ctx.group)
        d_rhs = (2.0 * B.from_probe_major(s_y, N) * g_iq.to(wd).reshape(1, c)).to(rhs.dtype) if ctx.needs_input_grad[5] else None
        return None, d_ls, d_os, d_ktt, d_noise, d_rhs, None, None, d_par


# ------------------------------------------------------------------------------------------------ modules
class IndexKernel(Kernel):
    """``gpytorch/kernels/index_kernel.py``: K_TT = B B^T + diag(v)."""

    def __init__(self, num_tasks: int, rank: int = 1, prior=None, var_constraint=None, **kwargs):
        if rank > num_tasks:
            raise RuntimeError("Cannot create a task covariance matrix larger than the number of tasks")
        super().__init__(**kwargs)
        self.register_parameter("covar_factor", torch.nn.Parameter(torch.randn(num_tasks, rank)))
        self.register_parameter("raw_var", torch.nn.Parameter(torch.randn(num_tasks)))
        self.register_constraint("raw_var", Positive() if var_constraint is None else var_constraint)
        if prior is not None:       # index_kernel.py:73-76: a prior on the task covariance matrix B B^T + diag(v) itself
            from .module import AttrGetter

            self.register_prior("IndexKernelPrior", prior, AttrGetter("covar_matrix"))

    def _eval_covar_matrix(self):
        return self.covar_matrix

    @property
    def var(self):
        return self._get_transformed("raw_var")

    @var.setter
    def var(self, value):
        self._set_transformed("raw_var", value)

    @property
    def covar_matrix(self):
        return self.covar_factor @ self.covar_factor.mT + torch.diag(self.var)  # index_kernel.py:91-99

    def __call__(self, i1, i2=None, diag=False, **params):
        return self.forward(i1, i1 if i2 is None else i2, diag=diag, **params)

    def forward(self, i1, i2, diag=False, **params):
        """index_kernel.py:101-112: the task covariance looked up at the task indices of the points (Hadamard multitask GPs:
        ``covar_x.mul(covar_i)``, see :mod:`gpytorch_amd.hadamard`)."""
        from .hadamard import IndexedTaskCovar

        op = IndexedTaskCovar(self.covar_matrix, i1, i2)
        return op.diagonal() if diag else op


class MultitaskKernel(Kernel):
    """``gpytorch/kernels/multitask_kernel.py:46-61``."""

    def __init__(self, data_covar_module, num_tasks: int, rank: int = 1, task_covar_prior=None, **kwargs):
        super().__init__(**kwargs)
        self.task_covar_module = IndexKernel(num_tasks=num_tasks, rank=rank, prior=task_covar_prior)
        self.data_covar_module = data_covar_module
        self.num_tasks = num_tasks

    def __call__(self, x1, x2=None, diag=False, **params):
        return self.forward(x1, x2, diag=diag, **params)

    def forward(self, x1, x2, diag=False, **params):
        covar_x = self.data_covar_module(x1, x2, **params)
        res = KroneckerFusedLinearOperator(covar_x, self.task_covar_module.covar_matrix)
        return res.diagonal() if diag else res

    def num_outputs_per_input(self, x1, x2):
        return self.num_tasks


class MultitaskMean(Mean):
    """``gpytorch/means/multitask_mean.py:41-45``: one base mean per task, stacked on the last dimension."""

    def __init__(self, base_means, num_tasks: int):
        super().__init__()
        if isinstance(base_means, Mean):
            base_means = [base_means]
        if len(base_means) == 1:
            import copy

            base_means = base_means + [copy.deepcopy(base_means[0]) for _ in range(num_tasks - 1)]
        if len(base_means) != num_tasks:
            raise RuntimeError("base_means should be a list of means of length either 1 or num_tasks")
        self.base_means = torch.nn.ModuleList(base_means)
        self.num_tasks = num_tasks

    def forward(self, x):
        return torch.cat([m(x).unsqueeze(-1) for m in self.base_means], dim=-1)


class MultitaskMultivariateNormal(MultivariateNormal):
    """An [n, T] event (leading batch dimensions allowed) over an nT x nT covariance (``multitask_multivariate_normal.py:14-297``).  ``interleaved``
    (the default, and the layout of every operator this package builds): row = i T + tau, data-major; otherwise row = tau n + i, task-major."""

    def __init__(self, mean: torch.Tensor, covariance_matrix, validate_args=False, interleaved=True):
        if mean.dim() < 2:
            raise RuntimeError("mean should be a matrix or a batch matrix (batch mode)")
        self._output_shape = mean.shape
        self._interleaved = interleaved
        flat = mean if interleaved else mean.mT
        super().__init__(flat.reshape(*mean.shape[:-2], -1), covariance_matrix)

    @property
    def event_shape(self):
        return self._output_shape[-2:]

    @property
    def batch_shape(self):
        return self._output_shape[:-2]

    @property
    def num_tasks(self):
        return self._output_shape[-1]

    def _to_event(self, flat):
        """[..., n T] in this distribution's flattening -> [..., n, T]."""
        n, T = self._output_shape[-2:]
        return flat.reshape(*flat.shape[:-1], n, T) if self._interleaved else flat.reshape(*flat.shape[:-1], T, n).mT

    def _to_flat(self, value):
        return (value if self._interleaved else value.mT).reshape(*value.shape[:-2], -1)

    @property
    def mean(self):
        return self._to_event(self.loc)

    @property
    def variance(self):
        return self._to_event(super().variance)

    def log_prob(self, value):
        return super().log_prob(self._to_flat(value))

    def rsample(self, sample_shape=torch.Size(), base_samples=None):
        if base_samples is not None:
            base_samples = self._to_flat(base_samples)
        return self._to_event(super().rsample(sample_shape, base_samples))

    sample = rsample

    def __add__(self, other):
        if isinstance(other, MultivariateNormal):
            return self.__class__(self.mean + other.mean, self.lazy_covariance_matrix + other.lazy_covariance_matrix, interleaved=self._interleaved)
        return self.__class__(self.mean + other, self._covar, interleaved=self._interleaved)

    def __mul__(self, other):
        if not isinstance(other, (int, float)):
            raise RuntimeError("Can only multiply by scalars")
        if other == 1:
            return self
        covar = self._covar.mul(other ** 2) if self.islazy else self._covar * (other ** 2)
        return self.__class__(self.mean * other, covar, interleaved=self._interleaved)

    def expand(self, batch_size):
        batch = torch.Size(batch_size)
        N = self.loc.shape[-1]
        return self.__class__(self.mean.expand(*batch, *self._output_shape[-2:]), self._rewrap(self.covariance_matrix.expand(*batch, N, N)), interleaved=self._interleaved)

    def __getitem__(self, idx):
        raise NotImplementedError("indexing a MultitaskMultivariateNormal: index its mean / covariance_matrix")

    # ---- constructors from single-output distributions (``:86-190``).  The tasks are INDEPENDENT here: the covariance is block diagonal.  It is
    # assembled densely (the reference, too, evaluates the member covariances for ``from_independent_mvns``); a batch of GPs that should stay on
    # the fused path is trained and queried as a batch model -- these constructors serve the layers that want an [n, T]-shaped result.
    @classmethod
    def from_batch_mvn(cls, batch_mvn, task_dim=-1):
        nb = len(batch_mvn.batch_shape)
        td = task_dim if task_dim >= 0 else nb + task_dim
        if td < 0 or td >= max(nb, 1) or nb == 0:
            raise ValueError(f"task_dim of {task_dim} is incompatible with MVN batch shape of {batch_mvn.batch_shape}")
        mean = batch_mvn.mean.movedim(td, -1)                                   # [*rest, n, T]
        blocks = batch_mvn.covariance_matrix.movedim(td, -3)                    # [*rest, T, n, n]
        T, n = blocks.shape[-3], blocks.shape[-1]
        full = torch.zeros(*blocks.shape[:-3], n, T, n, T, dtype=blocks.dtype, device=blocks.device)
        tau = torch.arange(T, device=blocks.device)
        full[..., :, tau, :, tau] = blocks.movedim(-3, 0)                       # entry (i, tau; j, tau) = K_tau[i, j]: interleaved block diagonal
        return cls(mean, full.reshape(*blocks.shape[:-3], n * T, n * T))

    @classmethod
    def from_independent_mvns(cls, mvns):
        if len(mvns) < 2:
            raise ValueError("Must provide at least 2 MVNs to form a MultitaskMultivariateNormal")
        if any(isinstance(m, MultitaskMultivariateNormal) for m in mvns):
            raise ValueError("Cannot accept MultitaskMultivariateNormals")
        if not all(m.batch_shape == mvns[0].batch_shape for m in mvns[1:]):
            batch = torch.broadcast_shapes(*(m.batch_shape for m in mvns))
            mvns = [m.expand(batch) for m in mvns]
        if not all(m.event_shape == mvns[0].event_shape for m in mvns[1:]):
            raise ValueError("All MultivariateNormals must have the same event shape")
        mean = torch.stack([m.mean for m in mvns], -1)
        covar = torch.stack([torch.block_diag(*bs) for bs in zip(*[m.covariance_matrix.reshape(-1, *m.covariance_matrix.shape[-2:]) for m in mvns])])
        N = covar.shape[-1]
        return cls(mean, covar.reshape(*mean.shape[:-2], N, N), interleaved=False)          # task-major, as the reference returns it

    @classmethod
    def from_repeated_mvn(cls, mvn, num_tasks):
        return cls.from_batch_mvn(mvn.expand(torch.Size([num_tasks]) + mvn.batch_shape), task_dim=0)

    def to_data_independent_dist(self, jitter_val=1e-4):
        """``:255-276``: one T x T distribution per data point (the task covariance at each point, cross-point covariances dropped)."""
        n, T = self._output_shape[-2:]
        full = self.covariance_matrix
        if self._interleaved:
            data = torch.arange(0, n * T, T, device=full.device).view(-1, 1, 1)
            task = torch.arange(T, device=full.device)
        else:
            data = torch.arange(n, device=full.device).view(-1, 1, 1)
            task = torch.arange(0, n * T, n, device=full.device)
        task_covars = full[..., data + task.unsqueeze(-2), data + task.unsqueeze(-1)]
        return MultivariateNormal(self.mean, task_covars + jitter_val * torch.eye(T, dtype=full.dtype, device=full.device))


class MultitaskGaussianLikelihood(_GaussianLikelihoodBase):
    """``gpytorch/likelihoods/multitask_gaussian_likelihood.py`` with rank = 0: noise I_n (x) (diag(task_noises) + noise I_T)."""

    def __init__(self, num_tasks: int, rank: int = 0, noise_constraint=None, has_global_noise=True, has_task_noise=True, **kwargs):
        if rank != 0:
            raise NotImplementedError("inter-task noise correlations (rank > 0) are not implemented")
        Module.__init__(self)
        self.num_tasks = num_tasks
        self.has_global_noise, self.has_task_noise = has_global_noise, has_task_noise
        if has_task_noise:
            self.register_parameter("raw_task_noises", torch.nn.Parameter(torch.zeros(num_tasks)))
            self.register_constraint("raw_task_noises", GreaterThan(1e-4) if noise_constraint is None else noise_constraint)
        if has_global_noise:
            self.register_parameter("raw_noise", torch.nn.Parameter(torch.zeros(1)))
            self.register_constraint("raw_noise", GreaterThan(1e-4) if noise_constraint is None else noise_constraint)

    @property
    def task_noises(self):
        return self._get_transformed("raw_task_noises")

    @task_noises.setter
    def task_noises(self, value):
        self._set_transformed("raw_task_noises", value)

    @property
    def noise(self):
        return self._get_transformed("raw_noise")

    @noise.setter
    def noise(self, value):
        self._set_transformed("raw_noise", value)

    def _task_noise_vector(self):
        dev = (self.raw_task_noises if self.has_task_noise else self.raw_noise).device
        out = torch.zeros(self.num_tasks, device=dev)
        if self.has_task_noise:
            out = out + self.task_noises
        if self.has_global_noise:
            out = out + self.noise
        return out

    @property
    def task_noise_covar(self):
        raise AttributeError("This likelihood holds diagonal task noises (rank 0): there is no low-rank task noise covariance")

    @task_noise_covar.setter
    def task_noise_covar(self, value):
        # multitask_gaussian_likelihood.py:262-270: a full task-noise matrix can only be set on a rank > 0 likelihood
        raise AttributeError("Cannot set non-diagonal task noises when covariance is diagonal.")

    def marginal(self, function_dist, *params, **kwargs):
        mean, covar = function_dist.mean, function_dist.lazy_covariance_matrix
        n = covar.shape[-1] // self.num_tasks
        if getattr(function_dist, "_interleaved", True):
            return function_dist.__class__(mean, covar + TaskNoiseDiagLinearOperator(self._task_noise_vector(), n))
        # task-major flattening (multitask_gaussian_likelihood.py:118-141 orders the Kronecker factors of the noise the same way)
        from .operators import DiagLinearOperator

        return function_dist.__class__(mean, covar + DiagLinearOperator(self._task_noise_vector().repeat_interleave(n)), interleaved=False)
