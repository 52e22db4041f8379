"""LinearOperator-protocol classes of the BBMM path.

The reference's operator algebra lives in the third-party ``linear_operator`` package; only the
members the ExactGP hot path touches are mirrored here, with the same names and argument meaning
(seam 2):

  ``_matmul / matmul / @``, ``_size / shape``, ``_transpose_nonbatch / mT``, ``diagonal``, ``to_dense``,
  ``evaluate_kernel``, ``__getitem__`` (row / column slices), ``mul`` (ScaleKernel), ``__add__`` /
  ``add_diagonal`` (likelihood noise), ``inv_quad_logdet``, ``inv_quad``, ``logdet``, ``solve``,
  ``pivoted_cholesky``, ``root_inv_decomposition``, ``detach``.

:class:`FusedKernelLinearOperator` plays the role of ``KernelLinearOperator`` as constructed at
``gpytorch/kernels/keops/rbf_kernel.py:48`` (matrix-free, K never formed) and
:class:`FusedKernelAddedDiagLinearOperator` that of ``AddedDiagLinearOperator`` built by
``_GaussianLikelihoodBase.marginal`` (``gpytorch/likelihoods/gaussian_likelihood.py:117-121``); their
hot methods run on the HIP kernels through :mod:`gpytorch_amd.functions`.
"""
from __future__ import annotations

import torch

from . import backend as B
from . import settings
from .functions import CholeskyInvQuadLogdetFn, InvQuadLogdetFn, KernelDenseFn, KernelMatmulFn, KernelSpec, SolveFn


class NotPSDError(RuntimeError):
    """Same role as ``linear_operator.utils.errors.NotPSDError``."""


class NanError(RuntimeError):
    """Same role as ``linear_operator.utils.errors.NanError``."""


def psd_safe_cholesky(A: torch.Tensor, jitter=None, max_tries=None, model_dtype=None) -> torch.Tensor:
    """``linear_operator.utils.cholesky.psd_safe_cholesky`` (third-party, restated): the lower Cholesky factor of ``A`` ([..., n, n]);
    when the plain factorisation fails, jitter ``settings.cholesky_jitter`` x 10^i (i = 0 .. ``settings.cholesky_max_tries`` - 1) is added to the diagonal of
    the FAILED batch members only, with a ``NumericalWarning`` per level; NaN input raises ``NanError``, exhaustion ``NotPSDError``
    (every small-n branch of the reference reaches its factor through it: ``LinearOperator.cholesky`` under ``inv_quad_logdet`` /
    ``solve`` / ``root_decomposition``).  The successful first attempt -- the normal case -- is exactly ``torch.linalg.cholesky``.
    ``model_dtype``: dtype whose jitter default applies when ``A`` was promoted for the factorisation (float32 models factorise in float64)."""
    L, info = torch.linalg.cholesky_ex(A)
    if not bool(info.any()):
        return L
    if bool(torch.isnan(A).any()):
        raise NanError(f"cholesky: {int(torch.isnan(A).sum())} of {A.numel()} elements of the {tuple(A.shape)} tensor are NaN.")
    import warnings

    from .linear_cg import NumericalWarning

    if jitter is None:
        jitter = settings.cholesky_jitter.value(model_dtype if model_dtype is not None else A.dtype)
    if max_tries is None:
        max_tries = settings.cholesky_max_tries.value()
    prev, new, Ap = 0.0, 0.0, A
    for i in range(max_tries):
        new = jitter * (10**i)
        add = (info > 0).to(A.dtype) * (new - prev)
        Ap = Ap + torch.diag_embed(add.reshape(*A.shape[:-2], 1).expand(*A.shape[:-1]))
        prev = new
        warnings.warn(f"A not p.d., added jitter of {new:.1e} to the diagonal", NumericalWarning)
        L, info = torch.linalg.cholesky_ex(Ap)
        if not bool(info.any()):
            return L
    raise NotPSDError(f"Matrix not positive definite after repeatedly adding jitter up to {new:.1e}.")


class LinearOperator:
    """Minimal protocol base (2-D, no batch dimensions: the fused kernels are non-batched).

    Subclasses written against ``linear_operator``'s own protocol (``gpytorch_amd.dropin``: every tensor goes to ``super().__init__`` so that
    ``representation()`` can list it; ``_getitem(row_index, col_index, *batch_indices)`` and ``_diagonal()`` instead of ``__getitem__`` /
    ``diagonal``) work on this base as they do on the real one: ``__init__`` keeps its arguments, ``__getitem__`` / ``diagonal`` route to the
    underscore methods where a subclass defines them."""

    def __init__(self, *args, **kwargs):
        self._args, self._kwargs = args, kwargs

    def representation(self):
        """The tensors this operator was built from (``LinearOperator.representation``: operators among the arguments contribute theirs)."""
        out = []
        for a in getattr(self, "_args", ()):
            if torch.is_tensor(a):
                out.append(a)
            elif isinstance(a, LinearOperator):
                out.extend(a.representation())
        return tuple(out)

    def _size(self) -> torch.Size:
        raise NotImplementedError

    def _matmul(self, rhs: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def _transpose_nonbatch(self) -> "LinearOperator":
        raise NotImplementedError

    # ---- generic API ----
    @property
    def shape(self):
        return self._size()

    def size(self, dim=None):
        s = self._size()
        return s if dim is None else s[dim]

    def dim(self):
        return len(self._size())

    @property
    def batch_shape(self):
        return torch.Size([])

    @property
    def is_square(self):
        return self.shape[-1] == self.shape[-2]

    def matmul(self, rhs):
        if isinstance(rhs, LinearOperator):
            return MatmulLinearOperator(self, rhs)
        squeeze = rhs.dim() == 1
        out = self._matmul(rhs.unsqueeze(-1) if squeeze else rhs)
        return out.squeeze(-1) if squeeze else out

    __matmul__ = matmul

    @property
    def mT(self):
        return self._transpose_nonbatch()

    def t(self):
        return self._transpose_nonbatch()

    def transpose(self, d0, d1):
        return self._transpose_nonbatch()

    def evaluate_kernel(self):
        return self

    def to_dense(self) -> torch.Tensor:
        n = self.shape[-1]
        return self._matmul(torch.eye(n, device=self.device, dtype=self.dtype))

    def diagonal(self, offset=0, dim1=-2, dim2=-1):
        if type(self)._diagonal is not LinearOperator._diagonal:    # a subclass on linear_operator's protocol
            return self._diagonal()
        return self.to_dense().diagonal()

    def _diagonal(self):
        return self.diagonal()

    def add_diagonal(self, diag: torch.Tensor):
        n = self.shape[-1]
        if diag.numel() == 1:
            return self + ConstantDiagLinearOperator(diag.reshape(1), n)
        return self + DiagLinearOperator(diag)

    def add_jitter(self, jitter_val=1e-3):
        return self.add_diagonal(torch.tensor([jitter_val], device=self.device, dtype=self.dtype))

    def __add__(self, other):
        if isinstance(other, ZeroLinearOperator):
            return self
        if isinstance(other, torch.Tensor):
            other = DenseLinearOperator(other)
        return SumLinearOperator(self, other)

    def mul(self, other):
        if isinstance(other, (int, float)):
            other = torch.tensor([float(other)], device=self.device, dtype=self.dtype)
        if isinstance(other, LinearOperator):  # elementwise product of two operators (e.g. data kernel o task covariance)
            if hasattr(other, "ktt") and isinstance(self, FusedKernelLinearOperator):
                return other.mul(self)
            return DenseLinearOperator(self.to_dense() * other.to_dense())
        return self._mul_constant(other)

    __mul__ = mul

    def _mul_constant(self, c):
        return DenseLinearOperator(self.to_dense() * c.reshape(()))

    def __getitem__(self, index):
        if hasattr(self, "_getitem"):    # a subclass on linear_operator's protocol: (row index, column index), both always present
            index = _strip_ellipsis(index)
            if not isinstance(index, tuple):
                index = (index, slice(None))
            if len(index) == 2:
                return self._getitem(index[0], index[1])
        return DenseLinearOperator(self.to_dense()[_strip_ellipsis(index)])

    def detach(self):
        return self

    def sum(self, dim=None):
        """``LinearOperator.sum``: row / column sums through one product with a ones vector (matrix-free)."""
        n, m = self.shape[-2], self.shape[-1]
        if dim is None:
            return self._matmul(torch.ones(m, 1, device=self.device, dtype=self.dtype)).sum()
        if dim in (-1, 1):
            return self._matmul(torch.ones(m, 1, device=self.device, dtype=self.dtype)).squeeze(-1)
        if dim in (-2, 0):
            return self._transpose_nonbatch()._matmul(torch.ones(n, 1, device=self.device, dtype=self.dtype)).squeeze(-1)
        raise ValueError(f"invalid dim {dim}")

    def numel(self):
        return self.shape[-1] * self.shape[-2]

    # ---- solves / determinants: dense Cholesky defaults (small operators only) ----
    def cholesky(self):
        return psd_safe_cholesky(self.to_dense().to(torch.float64), model_dtype=self.dtype)

    def solve(self, rhs: torch.Tensor, lhs=None) -> torch.Tensor:
        squeeze = rhs.dim() == 1
        r = rhs.unsqueeze(-1) if squeeze else rhs
        sol = torch.cholesky_solve(r.to(torch.float64), self.cholesky()).to(rhs.dtype)
        if lhs is not None:
            sol = lhs @ sol
        return sol.squeeze(-1) if squeeze else sol

    def inv_quad_logdet(self, inv_quad_rhs=None, logdet=False, reduce_inv_quad=True):
        Lc = self.cholesky()
        iq = None
        if inv_quad_rhs is not None:
            r = inv_quad_rhs.unsqueeze(-1) if inv_quad_rhs.dim() == 1 else inv_quad_rhs
            sol = torch.cholesky_solve(r.to(torch.float64), Lc)
            iq = (sol * r.to(torch.float64)).sum(-2).to(r.dtype)
            if reduce_inv_quad:
                iq = iq.sum(-1)
        ld = (2.0 * Lc.diagonal(dim1=-2, dim2=-1).log().sum(-1)).to(self.dtype) if logdet else None      # (one value per batch member)
        return iq, ld

    def inv_quad(self, inv_quad_rhs, reduce_inv_quad=True):
        return self.inv_quad_logdet(inv_quad_rhs, False, reduce_inv_quad)[0]

    def logdet(self):
        return self.inv_quad_logdet(None, True)[1]

    def root_decomposition(self, method=None):
        """``LinearOperator.root_decomposition``: R with R R^T ~= self.  "cholesky" for small operators, "lanczos" (rank
        ``max_root_decomposition_size``, through this operator's own matrix-free product) above ``max_cholesky_size`` or when
        ``settings.fast_pred_samples`` asks for the low-rank LOVE-style root."""
        n = self.shape[-1]
        method = check_root_method(method)
        if method is None:
            big = n > settings.max_cholesky_size.value() or settings.fast_pred_samples.on()
            method = "lanczos" if (big and self.device is not None and torch.device(self.device).type == "cuda") else "cholesky"
        if method == "cholesky":
            # (psd_safe_cholesky raises NotPSDError after its last jitter level: a failed factor is never handed back)
            Lc = psd_safe_cholesky(self.to_dense().to(torch.float64), model_dtype=self.dtype)
            return RootLinearOperator(Lc.to(self.dtype))
        if method == "symeig":
            return RootLinearOperator(_symeig_root(self.to_dense(), False, self.dtype))
        if method == "pivoted_cholesky":
            # low-rank root L with L L^T ~= self: the greedy factor itself (rank max_root_decomposition_size), row by row -- matrix-free wherever
            # the operator serves rows (the fused kernel operators do); the reference: RootLinearOperator(self.pivoted_cholesky(rank))
            from .bbmm import pivoted_cholesky_rows

            rank = min(settings.max_root_decomposition_size.value(), n)
            if type(self).pivoted_cholesky is not LinearOperator.pivoted_cholesky:
                try:
                    return RootLinearOperator(self.pivoted_cholesky(rank).to(self.dtype))
                except NotImplementedError:
                    pass
            lt = pivoted_cholesky_rows(self._row, self.diagonal().detach(), rank, settings.preconditioner_tolerance.value())
            return RootLinearOperator(lt.t().contiguous().to(self.dtype))
        from .lanczos import lanczos_tridiag, tridiag_to_diag

        def matvec(q_row):  # probe-major [1, ld] -> [1, ld]
            out = self._matmul(q_row[:, :n].t().to(self.dtype))
            res = torch.zeros_like(q_row)
            res[:, :n] = out.t().to(q_row.dtype)
            return res

        Q, T = lanczos_tridiag(None, None, None, settings.max_root_decomposition_size.value(), matvec=matvec, nvec=n, device=self.device)
        evals, evecs = tridiag_to_diag(T)
        w = (evecs * evals.clamp_min(0).sqrt().unsqueeze(-2)).to(device=Q.device, dtype=Q.dtype)      # V Lambda^1/2
        return RootLinearOperator((w.t() @ Q)[:, :n].t().contiguous().to(self.dtype))

    def _row(self, p: torch.Tensor) -> torch.Tensor:
        """Row p ([n]) of the matrix for a 1-element index tensor (``bbmm.pivoted_cholesky_rows``); dense default."""
        return self.to_dense()[int(p.reshape(-1)[0])]

    def pivoted_cholesky(self, rank, error_tol=None, return_pivots=False):
        """``LinearOperator.pivoted_cholesky`` (wrapper ``gpytorch/__init__.py:146-173``): the greedy rank-``rank`` factor L [n, rank], row by row."""
        from .bbmm import pivoted_cholesky_rows

        if return_pivots:
            raise NotImplementedError("pivoted_cholesky(return_pivots=True) on this operator")
        tol = settings.preconditioner_tolerance.value() if error_tol is None else error_tol
        return pivoted_cholesky_rows(self._row, self.diagonal().detach(), min(rank, self.shape[-1]), tol).t().contiguous()

    def zero_mean_mvn_samples(self, num_samples: int) -> torch.Tensor:
        """``LinearOperator.zero_mean_mvn_samples``: [num_samples, n] draws of N(0, self) through a root decomposition, or --
        ``settings.ciq_samples`` -- as K^{1/2} eps by contour-integral quadrature (no root is ever formed)."""
        if settings.ciq_samples.on():
            from .ciq import sqrt_matmul

            eps = torch.randn(self.shape[-1], num_samples, device=self.device, dtype=self.dtype)
            return sqrt_matmul(self, eps).t()
        root = self.root_decomposition().root
        eps = torch.randn(*root.shape[:-2], root.shape[-1], num_samples, device=root.device, dtype=root.dtype)
        return torch.movedim(root @ eps, -1, 0)   # [num_samples, ..., n]: batch operators keep their leading dimensions

    def root_inv_decomposition(self, initial_vectors=None, test_vectors=None, method=None):
        """Dense default (small operators): the Cholesky root of the inverse.  ``method="lanczos"`` on an operator without a matrix-free
        Lanczos of its own is an error, not a silent Cholesky."""
        method = check_root_method(method, inverse=True)
        if method == "lanczos":
            raise NotImplementedError(f"{type(self).__name__}.root_inv_decomposition(method='lanczos')")
        lanczos_vectors(initial_vectors, test_vectors, self.shape[-1], self.dtype)   # (shape errors as the reference raises them; a dense factor needs no start vector)
        if method == "symeig":
            return RootLinearOperator(_symeig_root(self.to_dense(), True, self.dtype))
        Lc = self.cholesky()
        inv_root = torch.linalg.solve_triangular(Lc, torch.eye(Lc.shape[-1], dtype=Lc.dtype, device=Lc.device), upper=False).mT
        return RootLinearOperator(inv_root.to(self.dtype))

    @property
    def requires_grad(self):
        return False


class DenseLinearOperator(LinearOperator):
    def __init__(self, tensor: torch.Tensor):
        self.tensor = tensor

    dtype = property(lambda self: self.tensor.dtype)
    device = property(lambda self: self.tensor.device)

    def _size(self):
        return self.tensor.shape

    def _matmul(self, rhs):
        return self.tensor @ rhs

    def _transpose_nonbatch(self):
        return DenseLinearOperator(self.tensor.mT)

    def to_dense(self):
        return self.tensor

    def diagonal(self, offset=0, dim1=-2, dim2=-1):
        return self.tensor.diagonal(dim1=-2, dim2=-1)

    def detach(self):
        return DenseLinearOperator(self.tensor.detach())


def to_linear_operator(obj):
    return obj if isinstance(obj, LinearOperator) else DenseLinearOperator(obj)


def to_dense(obj):
    return obj.to_dense() if isinstance(obj, LinearOperator) else obj


def check_root_method(method, inverse: bool = False):
    """``method`` of ``LinearOperator.root_decomposition`` / ``root_inv_decomposition`` (``gpytorch/__init__.py:176-216``): "cholesky", "lanczos",
    "symeig", and for the forward root also "pivoted_cholesky" -- or an error, never silently another method."""
    allowed = (None, "cholesky", "lanczos", "symeig") + (() if inverse else ("pivoted_cholesky",))
    if method not in allowed:
        raise NotImplementedError(f"root {'inverse ' if inverse else ''}decomposition method {method!r}: one of {allowed[1:]}")
    return method


def _symeig_root(dense: torch.Tensor, inverse: bool, dtype) -> torch.Tensor:
    """V Lambda^(+-1/2) from the dense symmetric eigendecomposition in float64 (``method="symeig"``); eigenvalues below ``tridiagonal_jitter`` relative to
    the largest are treated as the reference treats non-positive ones (clamped for the root, dropped for the inverse root)."""
    evals, evecs = torch.linalg.eigh(dense.to(torch.float64))
    floor = settings.tridiagonal_jitter.value() * float(evals.abs().max()) * 1e-6
    if inverse:
        keep = evals > floor
        return (evecs[..., keep] / evals[keep].sqrt()).to(dtype)
    return (evecs * evals.clamp_min(0.0).sqrt().unsqueeze(-2)).to(dtype)


def lanczos_vectors(initial_vectors, test_vectors, n: int, dtype):
    """The reference's ``initial_vectors`` [n] / [n, b] and ``test_vectors`` [n] / [n, c] as probe-major blocks ([b, ld], [c, ld]) or None;
    shape errors as ``LinearOperator.root_inv_decomposition`` raises them (RuntimeError on a length mismatch)."""
    out = []
    for name, v in (("initial_vectors", initial_vectors), ("test_vectors", test_vectors)):
        if v is None:
            out.append(None)
            continue
        v = v.unsqueeze(-1) if v.dim() == 1 else v
        if v.shape[-2] != n:
            raise RuntimeError(f"LinearOperator (size={n} x {n}) cannot be multiplied with {name} (size={tuple(v.shape)}).")
        if v.dim() != 2:
            # batch-shaped vectors ([*batch, n, k]: what the reference's root_inv_decomposition accepts for batch operators).  The operators here
            # are single matrices whose DENSE defaults need no start vector -- accepted and ignored there (advisor finding, round 5); the matrix-free
            # Lanczos paths of a single operator cannot use a batch of them
            out.append(None)
            continue
        out.append(B.to_probe_major(v, dtype))
    return out[0], out[1]


class ZeroLinearOperator(LinearOperator):
    def __init__(self, *sizes, dtype=torch.float32, device=None):
        self._sizes = torch.Size(sizes)
        self.dtype, self.device = dtype, device

    def _size(self):
        return self._sizes

    def _matmul(self, rhs):
        return torch.zeros(self._sizes[-2], rhs.shape[-1], dtype=rhs.dtype, device=rhs.device)

    def _transpose_nonbatch(self):
        return ZeroLinearOperator(self._sizes[-1], self._sizes[-2], dtype=self.dtype, device=self.device)

    def to_dense(self):
        return torch.zeros(*self._sizes, dtype=self.dtype, device=self.device)

    def diagonal(self, offset=0, dim1=-2, dim2=-1):
        return torch.zeros(self._sizes[-1], dtype=self.dtype, device=self.device)

    def __add__(self, other):
        return other


class DiagLinearOperator(LinearOperator):
    def __init__(self, diag: torch.Tensor):
        self._diag = diag

    @property
    def batch_shape(self):
        return self._diag.shape[:-1]

    dtype = property(lambda self: self._diag.dtype)
    device = property(lambda self: self._diag.device)

    def _size(self):
        n = self._diag.shape[-1]
        return torch.Size([*self._diag.shape[:-1], n, n])

    def _matmul(self, rhs):
        return self._diag.unsqueeze(-1) * rhs

    def _transpose_nonbatch(self):
        return self

    def diagonal(self, offset=0, dim1=-2, dim2=-1):
        return self._diag

    def to_dense(self):
        return torch.diag_embed(self._diag)

    def inv_quad_logdet(self, inv_quad_rhs=None, logdet=False, reduce_inv_quad=True):
        iq = None
        if inv_quad_rhs is not None:
            r = inv_quad_rhs.unsqueeze(-1) if inv_quad_rhs.dim() == 1 else inv_quad_rhs
            iq = (r.pow(2) / self._diag.unsqueeze(-1)).sum(-2)
            if reduce_inv_quad:
                iq = iq.sum(-1)
        return iq, (self._diag.log().sum(-1) if logdet else None)


class ConstantDiagLinearOperator(DiagLinearOperator):
    """``noise_models.py:92``: sigma^2 I as a 1-element tensor + size."""

    def __init__(self, diag_values: torch.Tensor, diag_shape: int):
        self.diag_values = diag_values
        self.diag_shape = diag_shape

    @property
    def _diag(self):
        return self.diag_values.expand(*self.diag_values.shape[:-1], self.diag_shape)

    @property
    def batch_shape(self):
        return self.diag_values.shape[:-1]


class FixedPlusConstantDiagLinearOperator(DiagLinearOperator):
    """diag(fixed) + c I with the two parts kept APART: the fixed per-point noise of ``FixedNoiseGaussianLikelihood`` plus its learned
    ``second_noise`` (``gaussian_likelihood.py:337-352``).  Folding them into one vector would cut the autograd path to the learned scalar
    (the fused operators carry the vector as a non-learnable epilogue diagonal and differentiate the scalar)."""

    def __init__(self, fixed: torch.Tensor, const: torch.Tensor):
        self.fixed = fixed                                                   # [*batch, n]
        self.const = const.reshape(-1)[:1] if const.numel() == 1 else const                                               # [1], or [*batch, 1] in batch mode

    @property
    def _diag(self):
        return self.fixed + self.const


def split_diag(other: DiagLinearOperator, device, dtype):
    """(scalar noise [1] -- the differentiable part --, fixed per-point vector or None) of a diagonal operator added to a kernel operator."""
    if isinstance(other, FixedPlusConstantDiagLinearOperator):
        return other.const, other.fixed
    if isinstance(other, ConstantDiagLinearOperator):
        return other.diag_values.reshape(-1)[:1], None
    d = other._diag
    if d.numel() > 0 and bool((d == d.reshape(-1)[0]).all()):
        return d.reshape(-1)[:1], None
    return torch.zeros(1, device=device, dtype=dtype), d


class RootLinearOperator(LinearOperator):
    def __init__(self, root: torch.Tensor):
        self.root = root

    dtype = property(lambda self: self.root.dtype)
    device = property(lambda self: self.root.device)

    def _size(self):
        n = self.root.shape[-2]
        return torch.Size([n, n])

    def root_decomposition(self, method=None):
        """An operator given AS its root is its own root decomposition, whatever the root's column count (the reference's
        ``RootLinearOperator.root_decomposition``; ``test/distributions/test_multivariate_normal.py:307-325`` samples through a 5 x 10 root)."""
        return self

    def _matmul(self, rhs):
        return self.root @ (self.root.mT @ rhs)

    def _transpose_nonbatch(self):
        return self

    def to_dense(self):
        return self.root @ self.root.mT

    def diagonal(self, offset=0, dim1=-2, dim2=-1):
        return self.root.pow(2).sum(-1)


class MatmulLinearOperator(LinearOperator):
    def __init__(self, left, right):
        self.left, self.right = to_linear_operator(left), to_linear_operator(right)

    dtype = property(lambda self: self.left.dtype)
    device = property(lambda self: self.left.device)

    def _size(self):
        return torch.Size([self.left.shape[-2], self.right.shape[-1]])

    def _matmul(self, rhs):
        return self.left._matmul(self.right._matmul(rhs))

    def _transpose_nonbatch(self):
        return MatmulLinearOperator(self.right._transpose_nonbatch(), self.left._transpose_nonbatch())

    def to_dense(self):
        return self.left.to_dense() @ self.right.to_dense()

    def diagonal(self, offset=0, dim1=-2, dim2=-1):
        return (self.left.to_dense() * self.right.to_dense().mT).sum(-1)


class SumLinearOperator(LinearOperator):
    def __init__(self, *ops):
        self.ops = [to_linear_operator(o) for o in ops]

    dtype = property(lambda self: self.ops[0].dtype)
    device = property(lambda self: self.ops[0].device)

    def _size(self):
        return self.ops[0].shape

    def _matmul(self, rhs):
        out = self.ops[0]._matmul(rhs)
        for o in self.ops[1:]:
            out = out + o._matmul(rhs)
        return out

    def _transpose_nonbatch(self):
        return SumLinearOperator(*[o._transpose_nonbatch() for o in self.ops])

    def evaluate_kernel(self):
        """A member that is a deferred kernel call (the reference's ``LazyEvaluatedKernelTensor``, ``lazy/lazy_evaluated_kernel_tensor.py:342-372``)
        is evaluated and the sum is re-formed with ``+``, so that a fused kernel operator takes its noise term into its own epilogue -- what
        ``AddedDiagLinearOperator.evaluate_kernel`` does in linear_operator (evaluated operator + diagonal)."""
        ev = [o.evaluate_kernel() for o in self.ops]
        if all(a is b for a, b in zip(ev, self.ops)):
            return self
        out = ev[0]
        for o in ev[1:]:
            out = out + o
        return out

    def _evaluated(self):
        """``self.evaluate_kernel()`` when that changes anything (a deferred kernel call among the members), else None."""
        ev = self.evaluate_kernel()
        return None if ev is self else ev

    # solves / decompositions of a sum with a deferred kernel member go to the EVALUATED sum (the fused operator with its noise), not to the
    # dense defaults of the base class
    def solve(self, rhs, lhs=None):
        ev = self._evaluated()
        return super().solve(rhs, lhs) if ev is None else ev.solve(rhs, lhs)

    def inv_quad_logdet(self, inv_quad_rhs=None, logdet=False, reduce_inv_quad=True):
        ev = self._evaluated()
        return super().inv_quad_logdet(inv_quad_rhs, logdet, reduce_inv_quad) if ev is None else ev.inv_quad_logdet(inv_quad_rhs, logdet, reduce_inv_quad)

    def root_decomposition(self, method=None):
        ev = self._evaluated()
        return super().root_decomposition(method) if ev is None else ev.root_decomposition(method)

    def root_inv_decomposition(self, initial_vectors=None, test_vectors=None, method=None):
        ev = self._evaluated()
        if ev is None:
            return super().root_inv_decomposition(initial_vectors, test_vectors, method)
        return ev.root_inv_decomposition(initial_vectors, test_vectors, method)

    def to_dense(self):
        out = self.ops[0].to_dense()
        for o in self.ops[1:]:
            out = out + o.to_dense()
        return out

    def diagonal(self, offset=0, dim1=-2, dim2=-1):
        out = self.ops[0].diagonal()
        for o in self.ops[1:]:
            out = out + o.diagonal()
        return out


# =================================================================================================
class FusedKernelLinearOperator(LinearOperator):
    """outputscale * k(x1, x2; lengthscale) -- matrix-free; every product runs on the fused HIP kernels.

    ``spec.shift`` is the Matern centring vector (``matern_kernel.py:94``: mean of x1 of the ORIGINAL
    kernel call; slices keep it)."""

    def __init__(self, x1, x2, spec: KernelSpec, lengthscale, outputscale=None):
        self.x1, self.x2 = x1, x2
        self.spec = spec
        self.lengthscale, self.outputscale = lengthscale, outputscale
        self._prep = None
        self._same = None

    dtype = property(lambda self: self.x1.dtype)
    device = property(lambda self: self.x1.device)

    @property
    def requires_grad(self):
        return bool(self.lengthscale.requires_grad or (self.outputscale is not None and self.outputscale.requires_grad)
                    or self.x1.requires_grad or self.x2.requires_grad or (self.spec.param is not None and self.spec.param.requires_grad))

    @property
    def square_same_inputs(self):
        """x1 and x2 are the same points (``torch.equal(x1, x2)`` in ``kernels/rbf_kernel.py:72``-style checks): identical
        storage, or -- compared once and cached -- identical values."""
        if self._same is None:
            x1, x2 = self.x1, self.x2
            self._same = x1 is x2 or (x1.shape == x2.shape and (x1.data_ptr() == x2.data_ptr() or bool(torch.equal(x1, x2))))
        return self._same

    def _size(self):
        return torch.Size([self.x1.shape[-2], self.x2.shape[-2]])

    def _os(self):
        return None if self.outputscale is None else self.outputscale.detach().reshape(-1)[:1].to(B.work_dtype(self.x1)).contiguous()

    def prepared(self):
        if self._prep is None:
            pv = self.spec.param_value()
            p1 = B.prep_points(self.spec.kind, self.x1, self.lengthscale, self.spec.shift, pv)
            p2 = p1 if self.square_same_inputs else B.prep_points(self.spec.kind, self.x2.to(self.x1.dtype), self.lengthscale, self.spec.shift, pv)
            self._prep = (p1, p2)
        return self._prep

    def _matmul(self, rhs):
        # (distinct x2 tensors keep their own identity so that autograd can hand each its gradient)
        x2 = self.x1 if (self.x2 is self.x1 or (self.square_same_inputs and not self.x2.requires_grad)) else self.x2
        return KernelMatmulFn.apply(self.x1, x2, self.lengthscale, self.outputscale, None, rhs, self.spec, self.spec.param)

    def _transpose_nonbatch(self):
        return FusedKernelLinearOperator(self.x2, self.x1, self.spec, self.lengthscale, self.outputscale)

    def _mul_constant(self, c):
        if c.numel() > 1:  # batch of output scales over one kernel matrix
            return BatchLinearOperator.replicate(self, c.shape if c.dim() > 0 else torch.Size([]))._mul_constant(c)
        os_ = c if self.outputscale is None else self.outputscale.reshape(()) * c.reshape(())
        return FusedKernelLinearOperator(self.x1, self.x2, self.spec, self.lengthscale, os_.reshape(1))

    def diagonal(self, offset=0, dim1=-2, dim2=-1):
        n = self.x1.shape[-2]
        if self.square_same_inputs:  # stationary kernel: k(0) = 1 (kernels/kernel.py:332-334)
            one = torch.ones(n, device=self.device, dtype=self.dtype)
            return one if self.outputscale is None else one * self.outputscale.reshape(())
        p1, p2 = self.prepared()
        return B.kernel_diag(p1, p2, self._os()).to(self.dtype)

    def to_dense(self):
        if torch.is_grad_enabled() and self.requires_grad:
            x2 = self.x1 if (self.x2 is self.x1 or (self.square_same_inputs and not self.x2.requires_grad)) else self.x2
            return KernelDenseFn.apply(self.x1, x2, self.lengthscale, self.outputscale, self.spec, self.spec.param)
        p1, p2 = self.prepared()
        return B.kernel_dense(p1, p2, self._os()).to(self.dtype)

    def __getitem__(self, index):
        index = _strip_ellipsis(index)
        if not isinstance(index, tuple):
            index = (index, slice(None))
        r, c = index
        if isinstance(r, int) or isinstance(c, int):
            return self.to_dense()[index]
        x1 = self.x1[r]
        x2 = x1 if (self.square_same_inputs and _same_index(r, c)) else self.x2[c]
        return FusedKernelLinearOperator(x1, x2, self.spec, self.lengthscale, self.outputscale)

    def rows(self, idx):
        p1, p2 = self.prepared()
        return B.kernel_rows(p1, idx, p2, self._os())

    def _row(self, p):
        return self.rows(p.reshape(1)).reshape(-1)

    def __add__(self, other):
        if isinstance(other, DiagLinearOperator) and other.batch_shape:
            return BatchLinearOperator.replicate(self, other.batch_shape) + other
        if isinstance(other, DiagLinearOperator) and self.is_square:
            # heteroskedastic fixed noise (FixedNoiseGaussianLikelihood) rides in the fused epilogue as a vector; the scalar part
            # (homoskedastic noise, or the learned second noise) stays a differentiable scalar
            noise, vec = split_diag(other, self.device, self.dtype)
            return FusedKernelAddedDiagLinearOperator(self, noise, noise_vec=vec)
        return super().__add__(other)

    def detach(self):
        x1 = self.x1.detach()
        x2 = x1 if self.x2 is self.x1 else self.x2.detach()
        return FusedKernelLinearOperator(
            x1, x2, self.spec, self.lengthscale.detach(), None if self.outputscale is None else self.outputscale.detach()
        )

    # ---- solves on the noise-free kernel matrix itself (``kernel(x, x).solve(rhs)``,
    # test/lazy/test_lazy_evaluated_kernel_tensor.py:69-113): the same BBMM machinery with a zero diagonal
    def _with_zero_diag(self):
        if not (self.is_square and self.square_same_inputs):
            raise RuntimeError("solve / inv_quad_logdet need a square kernel matrix k(x, x)")
        return FusedKernelAddedDiagLinearOperator(self, torch.zeros(1, device=self.device, dtype=self.dtype))

    def solve(self, rhs, lhs=None):
        return self._with_zero_diag().solve(rhs, lhs)

    def inv_quad_logdet(self, inv_quad_rhs=None, logdet=False, reduce_inv_quad=True):
        return self._with_zero_diag().inv_quad_logdet(inv_quad_rhs, logdet, reduce_inv_quad)

    def root_inv_decomposition(self, initial_vectors=None, test_vectors=None, method=None):
        return self._with_zero_diag().root_inv_decomposition(initial_vectors, test_vectors, method)

    def pivoted_cholesky(self, rank, error_tol=None, return_pivots=False):
        """``LinearOperator.pivoted_cholesky`` (wrapper ``gpytorch/__init__.py:146-173``): returns L (n x m)."""
        p1, _ = self.prepared()
        tol = settings.preconditioner_tolerance.value() if error_tol is None else error_tol
        Lt, piv, m = B.pivoted_cholesky(p1, self._os(), rank, tol)
        L = Lt.t().contiguous().to(self.dtype)
        return (L, piv) if return_pivots else L


def _strip_ellipsis(index):
    """``op[..., r, c]`` on a non-batch operator is ``op[r, c]``."""
    if isinstance(index, tuple) and len(index) > 0 and index[0] is Ellipsis:
        return index[1:] if len(index) > 2 else (index[1] if len(index) == 2 else slice(None))
    return index


def _same_index(r, c):
    if isinstance(r, slice) and isinstance(c, slice):
        return (r.start, r.stop, r.step) == (c.start, c.stop, c.step)
    return r is c


class FusedKernelAddedDiagLinearOperator(LinearOperator):
    """K_hat = outputscale * k(x, x) + noise * I  (constant diagonal) -- the operator the MLL and the
    prediction caches solve with.  ``bbmm_opts`` forwards probe / sharding options to the solver."""

    def __init__(self, kernel_op: FusedKernelLinearOperator, noise: torch.Tensor, bbmm_opts: dict | None = None, noise_vec=None):
        self.kernel_op = kernel_op
        self.noise = noise.reshape(-1)[:1]
        self.noise_vec = noise_vec  # optional fixed per-point diagonal [n] (added on top of the scalar noise)
        self.bbmm_opts = {} if bbmm_opts is None else bbmm_opts
        self._cache = {}

    def _dvec(self):
        if self.noise_vec is None:
            return None
        if "dvec" not in self._cache:
            n = self.shape[-1]
            wd = B.work_dtype(self.kernel_op.x1)
            dv = torch.zeros(B.round_up(n, 4), device=self.device, dtype=wd)
            dv[:n] = self.noise_vec.detach().to(wd)
            self._cache["dvec"] = dv
        return self._cache["dvec"]

    def _spec(self):
        k = self.kernel_op
        return k.spec if self.noise_vec is None else k.spec.with_dvec(self._dvec())

    dtype = property(lambda self: self.kernel_op.dtype)
    device = property(lambda self: self.kernel_op.device)

    @property
    def requires_grad(self):
        return self.kernel_op.requires_grad or self.noise.requires_grad

    def _size(self):
        return self.kernel_op._size()

    def _matmul(self, rhs):
        k = self.kernel_op
        return KernelMatmulFn.apply(k.x1, k.x1, k.lengthscale, k.outputscale, self.noise, rhs, self._spec(), k.spec.param)

    def _transpose_nonbatch(self):
        return self

    def diagonal(self, offset=0, dim1=-2, dim2=-1):
        d = self.kernel_op.diagonal() + self.noise.reshape(())
        return d if self.noise_vec is None else d + self.noise_vec

    def to_dense(self):
        K = self.kernel_op.to_dense()
        K = K + self.noise.reshape(()) * torch.eye(K.shape[-1], device=K.device, dtype=K.dtype)
        return K if self.noise_vec is None else K + torch.diag(self.noise_vec.to(K.dtype))

    def __add__(self, other):
        if isinstance(other, DiagLinearOperator) and not other.batch_shape:
            noise, vec = split_diag(other, self.device, self.dtype)
            nv = self.noise_vec if vec is None else (vec if self.noise_vec is None else self.noise_vec + vec)
            return FusedKernelAddedDiagLinearOperator(self.kernel_op, self.noise + noise, self.bbmm_opts, nv)
        return super().__add__(other)

    def detach(self):
        return FusedKernelAddedDiagLinearOperator(self.kernel_op.detach(), self.noise.detach(), self.bbmm_opts, self.noise_vec)

    def restrict(self, idx: torch.Tensor):
        """K_hat[idx][:, idx] as a fused operator over the selected points (``observation_nan_policy("mask")``:
        the reference wraps the operator in a ``MaskedLinearOperator``, exact_marginal_log_likelihood.py:68-77)."""
        nv = None if self.noise_vec is None else self.noise_vec[idx]
        return FusedKernelAddedDiagLinearOperator(self.kernel_op[idx, idx], self.noise, self.bbmm_opts, nv)

    def _use_cholesky(self, flag) -> bool:
        return flag.off() or self.shape[-1] <= settings.max_cholesky_size.value()

    def _nz(self):
        return self.noise.detach().reshape(-1)[:1].to(B.work_dtype(self.kernel_op.x1)).contiguous()

    # ---- A.1 dispatch ----
    def inv_quad_logdet(self, inv_quad_rhs=None, logdet=False, reduce_inv_quad=True):
        k = self.kernel_op
        n = self.shape[-1]
        if inv_quad_rhs is None:
            inv_quad_rhs = torch.zeros(n, 0, device=self.device, dtype=self.dtype)
        rhs = inv_quad_rhs.unsqueeze(-1) if inv_quad_rhs.dim() == 1 else inv_quad_rhs
        if self._use_cholesky(settings.fast_computations.log_prob):
            iq, ld = CholeskyInvQuadLogdetFn.apply(k.x1, k.lengthscale, k.outputscale, self.noise, rhs, self._spec(), k.spec.param)
        else:
            if rhs.shape[-1] == 0:
                rhs = torch.zeros(n, 1, device=self.device, dtype=self.dtype)
                drop = True
            else:
                drop = False
            iq, ld = InvQuadLogdetFn.apply(k.x1, k.lengthscale, k.outputscale, self.noise, rhs, self._spec(), self._iql_opts(), k.spec.param)
            if drop:
                iq = iq[:0]
        if reduce_inv_quad:
            iq = iq.sum(-1)
        return iq, (ld if logdet else None)

    def _iql_opts(self) -> dict:
        """Solver options of this evaluation: ``bbmm_opts`` completed with the ``settings.sharding`` probe group
        (each rank then draws its share of ``num_trace_samples`` from a rank-specific generator)."""
        opts = self.bbmm_opts
        # the probe group and the row group of THIS evaluation: the explicit groups of the settings.sharding scope, or -- sharding("auto") -- the
        # P x R grid the cost model picks from (n, probes); rows can be sharded for the single fused float32 kernel with a constant diagonal only
        rows_ok = self.noise_vec is None and self.kernel_op.prepared()[0].fused
        # (an EXPLICIT mll_row_group on an operator whose rows cannot be sharded still raises in bbmm.inv_quad_logdet_forward: never silently replicated)
        pg_auto, rg = settings.sharding.mll_groups(self.shape[-1], settings.num_trace_samples.value(), allow_rows=rows_ok or not settings.sharding.is_auto())
        if rg is not None and "row_group" not in opts and torch.distributed.get_world_size(rg) > 1:
            opts = dict(opts, row_group=rg)   # two-dimensional split (probe groups x row blocks), bbmm.inv_quad_logdet_forward
        group = opts.get("group", pg_auto)
        if group is None or "group" in opts or torch.distributed.get_world_size(group) == 1:
            return opts
        from .distributed import probe_shard

        world, rank = torch.distributed.get_world_size(group), torch.distributed.get_rank(group)
        t_total = settings.num_trace_samples.value()
        a, b = probe_shard(t_total, world, rank)
        if b - a < 1:
            raise ValueError(f"probe sharding needs num_trace_samples >= world size ({t_total} < {world})")
        opts = dict(opts, group=group, num_probes=b - a, t_total=t_total)
        if "generator" not in opts and opts.get("probes") is None:
            opts["generator"] = settings.sharding.rank_generator(group, self.device)
        return opts

    def _preconditioner(self):
        """``AddedDiagLinearOperator._preconditioner``: (closure on [n, c] tensors, None, logdet) or Nones."""
        from .bbmm import build_preconditioner

        if "precond" not in self._cache:
            p1, _ = self.kernel_op.prepared()
            self._cache["precond"] = build_preconditioner(p1, self.kernel_op._os(), self._nz(), dvec=self._dvec())
        pre = self._cache["precond"]
        if pre is None:
            return None, None, None

        def closure(v):
            vt = B.to_probe_major(v, B.work_dtype(self.kernel_op.x1))
            return B.from_probe_major(pre.apply_(vt, torch.zeros_like(vt)), v.shape[-2]).to(v.dtype)

        return closure, None, pre.logdet

    def solve(self, rhs: torch.Tensor, lhs=None) -> torch.Tensor:
        """K_hat^-1 rhs (``exact_prediction_strategies.py:286,444``).  Differentiable (``SolveFn``) when gradients are
        enabled and the operator or the right-hand side requires them; prediction caches normally run detached
        (``settings.detach_test_caches``) and take the plain path."""
        from .bbmm import build_preconditioner
        from .linear_cg import linear_cg

        squeeze = rhs.dim() == 1
        r = rhs.unsqueeze(-1) if squeeze else rhs
        if self._use_cholesky(settings.fast_computations.solves):
            sol = torch.cholesky_solve(r.detach().to(torch.float64), psd_safe_cholesky(self.to_dense().detach().to(torch.float64), model_dtype=self.dtype)).to(rhs.dtype)
        elif self._row_shard() is not None:
            sol = self._solve_row_sharded(r).to(rhs.dtype)
        elif torch.is_grad_enabled() and (self.requires_grad or r.requires_grad) and self.kernel_op.prepared()[0].fused:
            k = self.kernel_op
            self._preconditioner()
            sol = SolveFn.apply(k.x1, k.lengthscale, k.outputscale, self.noise, r, self._spec(), self._cache["precond"],
                                settings.cg_tolerance.value(), k.spec.param)
        else:
            p1, _ = self.kernel_op.prepared()
            if "precond" not in self._cache:
                self._cache["precond"] = build_preconditioner(p1, self.kernel_op._os(), self._nz(), dvec=self._dvec())
            sol_t, info = linear_cg(
                p1, self.kernel_op._os(), self._nz(), B.to_probe_major(r.detach(), p1.dtype), n_tridiag=0,
                tolerance=settings.cg_tolerance.value(), preconditioner=self._cache["precond"], dvec=self._dvec(),
            )
            self._cache["last_cg_info"] = info
            if settings.rhs_refinement.on() and p1.fused and sol_t.dtype == torch.float32:
                from .bbmm import refine_solves_   # (any number of columns since round 5: the n_test-column solve of the exact predictive variance too --
                #                                     one fused float64 product with all columns + one more float32 solve of the residuals)

                refine_solves_(p1, self.kernel_op._os(), self._nz(), B.to_probe_major(r.detach(), p1.dtype), sol_t, settings.cg_tolerance.value(), None,
                               self._cache["precond"], self._dvec())
            sol = B.from_probe_major(sol_t, self.shape[-1]).to(rhs.dtype)
        if lhs is not None:
            sol = lhs @ sol
        return sol.squeeze(-1) if squeeze else sol

    def float64_product_available(self) -> bool:
        """A fused float64 product exists for this operator (float32 model, d <= 16: ``csrc/kv_f64.hpp``)."""
        p1, _ = self.kernel_op.prepared()
        return bool(p1.fused and p1.dp <= B.FUSED_F64_MAX_DP)

    def matmul_float64(self, rhs: torch.Tensor) -> torch.Tensor:
        """K_hat @ rhs ([n, c] -> [n, c]) with the kernel entries, the contraction and the result in FLOAT64 on the prepared points of the float32
        path (``bbmm.matvec64``): the residual / ``A X`` product of the mixed-precision corrections (``settings.rhs_refinement``,
        ``bbmm.variational_inv_quad``).  ``None`` where no fused float64 kernel applies (d > 16)."""
        from .bbmm import matvec64

        p1, _ = self.kernel_op.prepared()
        if not self.float64_product_available():
            return None
        mv = matvec64(p1, self.kernel_op._os(), self._nz(), self._dvec())
        return B.from_probe_major(mv(B.to_probe_major(rhs.detach(), torch.float64)), self.shape[-1])

    # ---- row-sharded small-t solves (settings.sharding(row_group=...) or bbmm_opts["row_group"]) ----
    def _row_shard(self):
        group = self.bbmm_opts.get("row_group", settings.sharding.row_group())
        if group is None or torch.distributed.get_world_size(group) == 1:
            return None
        p1, _ = self.kernel_op.prepared()
        if not p1.fused:
            return None  # generic path (float64 / d > 16): replicated solve
        if "row_shard" not in self._cache:
            from .distributed import RowShard

            self._cache["row_shard"] = RowShard(p1, group)
        return self._cache["row_shard"]

    def _dvec_local(self, rs):
        dv = self._dvec()
        return None if dv is None else rs.local(dv.unsqueeze(0))[0]

    def _solve_row_sharded(self, r: torch.Tensor) -> torch.Tensor:
        """K_hat^-1 r with every rank owning a block of rows; returns the FULL solution [n, c] on every rank."""
        from .linear_cg import linear_cg

        rs = self._row_shard()
        n = self.shape[-1]
        rhs_loc = rs.local(B.to_probe_major(r.detach()))
        self._preconditioner()  # the reference default has one (max_preconditioner_size = 15): built replicated, applied row-sharded
        os_, nz, dvl = self.kernel_op._os(), self._nz(), self._dvec_local(rs)

        def cg32(rt):
            return linear_cg(None, os_, nz, rt, n_tridiag=0, tolerance=settings.cg_tolerance.value(),
                             dvec=dvl, row_shard=rs, preconditioner=self._cache["precond"])

        sol_loc, info = cg32(rhs_loc)
        self._cache["last_cg_info"] = info
        if settings.rhs_refinement.on() and sol_loc.dtype == torch.float32 and self.float64_product_available():
            # mixed-precision refinement of the row-sharded solve (round 6): the float64 residual of this rank's rows from ONE rectangular fused
            # float64 product (local rows x all columns, the search directions all-gathered in float64), one more sharded float32 solve of it
            from .bbmm import refine_with_

            x_loc64 = B.PreparedPoints(rs.x_loc.xp.to(torch.float64), rs.x_loc.n, rs.x_loc.d, rs.x_loc.dp, rs.x_loc.kind, rs.x_loc.param)
            x_all64 = B.PreparedPoints(rs.x_all.xp.to(torch.float64), rs.x_all.n, rs.x_all.d, rs.x_all.dp, rs.x_all.kind, rs.x_all.param)
            os64 = None if os_ is None else os_.detach().to(torch.float64)
            nz64 = None if nz is None else nz.detach().to(torch.float64)
            dv64 = None if dvl is None else dvl.to(torch.float64)

            def mv64(a64):
                out = torch.empty_like(a64)
                for c0 in range(0, a64.shape[0], 80):
                    blk = a64[c0 : c0 + 80].contiguous()
                    out[c0 : c0 + 80] = B.kv(x_loc64, x_all64, rs.gather(blk), scale=os64, dscale=nz64, vd=blk, dvec=dv64)
                return out

            def solve32(res):
                d_, inf = cg32(res)
                return d_, inf.iterations

            refine_with_(rhs_loc.to(torch.float64), sol_loc, mv64, solve32, settings.rhs_refinement.steps)
        return B.from_probe_major(rs.gather(sol_loc), n)

    def _root_inv_row_sharded(self, init_t):
        from .lanczos import root_inv_decomposition

        rs = self._row_shard()
        n = self.shape[-1]
        os_, nz, dvl = self.kernel_op._os(), self._nz(), self._dvec_local(rs)
        if init_t is None:
            # the same start vector on every rank: drawn at full length (as the single-process path does), rank 0's wins
            init_t = torch.zeros(1, B.round_up(n, 4), device=self.device)
            init_t[:, :n] = torch.randn(1, n, device=self.device, generator=self.bbmm_opts.get("generator"))
            rs.broadcast(init_t)
        rt_loc = root_inv_decomposition(None, None, None, init_vec_t=rs.local(init_t), nvec=rs.n_loc, device=self.device,
                                        matvec=lambda q: rs.kv_local(q, scale=os_, dscale=nz, dvec_loc=dvl), reduce=rs.allreduce, n_global=n)
        return rs.gather(rt_loc)

    def root_inv_decomposition(self, initial_vectors=None, test_vectors=None, method=None):
        """Lanczos root-inverse (``exact_prediction_strategies.py:271``) -> RootLinearOperator(n x m).  ``initial_vectors`` [n, b] /
        ``test_vectors`` [n, c] as documented at ``gpytorch/__init__.py:190-216``: b recurrences in lock-step (ONE b-column product per
        step), the decomposition that solves the test vectors best is returned."""
        from .lanczos import root_inv_decomposition

        method = check_root_method(method, inverse=True)
        if method in ("cholesky", "symeig") or (method is None and self._use_cholesky(settings.fast_computations.covar_root_decomposition)):
            return super().root_inv_decomposition(method=method)      # dense factorisations of a small operator (base class)
        n = self.shape[-1]
        init_t, test_t = lanczos_vectors(initial_vectors, test_vectors, n, B.work_dtype(self.kernel_op.x1))
        if self._row_shard() is not None:
            if init_t is not None and init_t.shape[0] > 1:
                raise NotImplementedError("root_inv_decomposition: several initial_vectors on a row-sharded operator")
            rt = self._root_inv_row_sharded(None if init_t is None else init_t.to(torch.float32))
            return RootLinearOperator(B.from_probe_major(rt, n).to(self.dtype))
        p1, _ = self.kernel_op.prepared()
        rt = root_inv_decomposition(p1, self.kernel_op._os(), self._nz(), init_vec_t=init_t, test_vec_t=test_t,
                                    generator=self.bbmm_opts.get("generator"), dvec=self._dvec())
        return RootLinearOperator(B.from_probe_major(rt, n).to(self.dtype))

    def pivoted_cholesky(self, rank, error_tol=None, return_pivots=False):
        raise NotImplementedError("pivoted_cholesky is defined on the noise-free kernel operator (self.kernel_op)")

    def _row(self, p):
        """Row p of K_hat: the fused kernel row + the diagonal entry (``root_decomposition(method="pivoted_cholesky")``: matrix-free)."""
        row = self.kernel_op._row(p).clone()
        i = int(p.reshape(-1)[0])
        row[i] += self.noise.detach().reshape(()).to(row.dtype) + (0.0 if self.noise_vec is None else self.noise_vec.detach()[i].to(row.dtype))
        return row

    # ---- both prediction caches from ONE sequence of two-column products ------------------------------------------------
    def can_fuse_caches(self) -> bool:
        """The mean-cache solve (one right-hand side) and the LOVE Lanczos run can share their kernel products: float32 fused
        kernels, CG branch for both, no row sharding."""
        p1, _ = self.kernel_op.prepared()
        return (p1.fused and not self._use_cholesky(settings.fast_computations.solves)
                and not self._use_cholesky(settings.fast_computations.covar_root_decomposition) and self._row_shard() is None
                and self._fused_lanczos_ok())

    def _fused_lanczos_ok(self) -> bool:
        from .lanczos import block_size_for

        rank = settings.max_root_decomposition_size.value()
        return rank <= 512 or block_size_for(self.shape[-1], rank) > 1     # (gpamd_lanczos_* kernels: k <= 512; the block form has no limit)

    def solve_and_root_inv(self, rhs: torch.Tensor):
        """(K_hat^-1 rhs, root of K_hat^-1) -- the mean cache and the LOVE covariance cache of
        ``exact_prediction_strategies.py:267-321`` -- computed TOGETHER: every mBCG iteration and every Lanczos step needs one
        product with K_hat, each is bound by kernel GENERATION (one v_exp_f32 per pair), so the CG direction and the Lanczos
        vector ride through ONE two-column launch (23.8 ms instead of 2 x 19.1 ms at n = 500 000).  The algorithms themselves
        are untouched (``linear_cg`` through its ``kv_partials`` hook, ``lanczos_steps`` as a coroutine)."""
        import ctypes as C

        from ._lib import check, lib
        from .lanczos import block_lanczos_steps, block_size_for, lanczos_steps, root_from_tridiag
        from .linear_cg import linear_cg

        p1, _ = self.kernel_op.prepared()
        n, dev = p1.n, self.device
        ld = B.round_up(n, 4)
        os_, nz, dv = self.kernel_op._os(), self._nz(), self._dvec()
        self._preconditioner()
        L, st = lib(), B._stream(dev)
        rank = settings.max_root_decomposition_size.value()
        gen = self.bbmm_opts.get("generator")
        nb = block_size_for(n, rank)          # rows per Lanczos product: 1 = the reference's recurrence, > 1 = block Lanczos
        if nb > 1:
            init = torch.zeros(nb, ld, device=dev, dtype=torch.float32)
            init[:, :n] = torch.randn(nb, n, device=dev, generator=gen, dtype=torch.float32)
            steps = block_lanczos_steps(n, dev, max(1, min(rank, n) // nb), init)
        else:
            steps = lanczos_steps(n, dev, rank, generator=gen)
        flags = B.kv_flags(p1, p1, 1 + nb)
        sorted_rows = B.rows_sorted(p1, p1, flags)
        S, jc, wsn = B.kv_plan(p1.kind, n, n, p1.d, 1 + nb, flags, ld)
        P = B.workspace(dev, wsn)
        W = torch.zeros(1 + nb, ld, device=dev, dtype=torch.float32)
        wl = torch.zeros(nb, ld, device=dev, dtype=torch.float32)
        state = {"q": next(steps), "result": None}

        def feed(w):
            try:
                state["q"] = steps.send(w)
            except StopIteration as done:
                state["result"] = done.value

        def hook(dt):
            if state["result"] is not None:   # Lanczos finished first: plain one-column products from here on
                if sorted_rows:
                    out1 = B.kv(p1, p1, dt)
                    return out1, 1, out1.stride(0)
                S1, jc1, ws1 = B.kv_plan(p1.kind, n, n, p1.d, 1, B.kv_flags(p1, p1, 1), ld)
                P1 = B.workspace(dev, wsn + ws1)[wsn:]
                check(L.gpamd_kv_partials_f32(*B.kind_args(p1), B._ptr(p1.xp), n, B._ptr(p1.xp), n, p1.d, None, B._ptr(dt), ld, 1, B._ptr(P1), ld,
                                              S1, jc1, B.kv_flags(p1, p1, 1), None, st), "kv_partials")
                return P1, S1, ld
            q = state["q"]
            W[0].copy_(dt[0])
            W[1:].copy_(q)
            if sorted_rows or nb > 1:
                # (block-centred Gram expansion -- wide clouds, backend.gram_mode 2 -- or a Lanczos BLOCK: B.kv sums the slabs and takes
                # Hilbert-ordered output rows back; the CG column is handed on as a single finished slab)
                both = B.kv(p1, p1, W)
                wl.copy_(both[1:])
                if os_ is not None:
                    wl.mul_(os_.reshape(()))
                wl[:, :n].addcmul_(q[:, :n], (nz.reshape(()) + (dv[:n] if dv is not None else 0.0)).expand(n))
                feed(wl)
                return both[0:1], 1, both.stride(0)
            check(L.gpamd_kv_partials_f32(*B.kind_args(p1), B._ptr(p1.xp), n, B._ptr(p1.xp), n, p1.d, None, B._ptr(W), ld, 2, B._ptr(P), ld, S, jc,
                                          flags, None, st), "kv_partials")
            # Lanczos column: slab row 1 of every split, rows 2 ld apart -> "t = 1 with ldp = 2 ld" for the reduction
            p_col1 = C.c_void_p(P.data_ptr() + 4 * ld)
            check(L.gpamd_kv_reduce_f32(p_col1, S, 2 * ld, 1, n, B._ptr(os_), B._ptr(nz), B._ptr(dv), B._ptr(q), q.stride(0), B._ptr(wl), ld, None, st),
                  "kv_reduce")
            feed(wl)
            return P, S, 2 * ld               # the CG column: slab row 0

        r = rhs.unsqueeze(-1) if rhs.dim() == 1 else rhs
        sol_t, info = linear_cg(p1, os_, nz, B.to_probe_major(r.detach(), p1.dtype), n_tridiag=0, tolerance=settings.cg_tolerance.value(),
                                preconditioner=self._cache["precond"], dvec=dv, kv_partials=hook)
        self._cache["last_cg_info"] = info
        if settings.rhs_refinement.on() and p1.fused and sol_t.dtype == torch.float32:
            from .bbmm import refine_solves_

            refine_solves_(p1, os_, nz, B.to_probe_major(r.detach(), p1.dtype), sol_t, settings.cg_tolerance.value(), None, self._cache["precond"], dv)
        while state["result"] is None:        # CG finished first: the remaining Lanczos steps on their own products
            q = state["q"]
            feed(B.kv(p1, p1, q, scale=os_, dscale=nz, vd=q, dvec=dv))
        Q, T = state["result"]
        root = B.from_probe_major(root_from_tridiag(Q, T), n).to(self.dtype)
        sol = B.from_probe_major(sol_t, n).to(rhs.dtype)
        return (sol.squeeze(-1) if rhs.dim() == 1 else sol), RootLinearOperator(root)


# =================================================================================================
class BatchLinearOperator(LinearOperator):
    """A batch of independent operators (leading ``batch_shape`` dimensions; the reference's batch mode,
    ``kernels/kernel.py:163-208`` ``batch_shape`` and ``test/examples/test_batch_gp_regression.py``).

    The fused kernels work on one point cloud at a time; a batch is a launch plan over its members: every method maps over
    ``self.ops`` (row-major over ``batch_shape``) and stacks.  Members of a batch GP are each as large as a single GP, so every
    launch still fills the chip; hyper-parameters with batch shape are sliced per member and autograd scatters the gradients
    back into the batched parameter."""

    def __init__(self, ops, batch_shape):
        self.ops = list(ops)
        self._batch_shape = torch.Size(batch_shape)
        assert len(self.ops) == max(1, self._batch_shape.numel())

    @classmethod
    def replicate(cls, op, batch_shape):
        batch_shape = torch.Size(batch_shape)
        return cls([op] * max(1, batch_shape.numel()), batch_shape)

    dtype = property(lambda self: self.ops[0].dtype)
    device = property(lambda self: self.ops[0].device)

    @property
    def batch_shape(self):
        return self._batch_shape

    def restrict(self, idx: torch.Tensor):
        """Every member restricted to the same subset of its points (``observation_nan_policy("mask")``: a point counts as observed only if every
        batch member observes it, settings.py:428-440)."""
        return BatchLinearOperator([op.restrict(idx) if hasattr(op, "restrict") else op[idx][:, idx] for op in self.ops], self._batch_shape)

    def _size(self):
        return torch.Size([*self._batch_shape, *self.ops[0].shape[-2:]])

    @property
    def requires_grad(self):
        return any(o.requires_grad for o in self.ops)

    # ---- helpers
    def _map(self, fn):
        return BatchLinearOperator([fn(o) for o in self.ops], self._batch_shape)

    def _stack(self, outs):
        out = torch.stack(list(outs), dim=0)
        return out.reshape(*self._batch_shape, *out.shape[1:])

    def _split(self, t: torch.Tensor, event_dims: int):
        """Broadcast ``t`` ([..., *event]) against the batch shape and return one slice per member."""
        ev = t.shape[t.dim() - event_dims :]
        tb = t.expand(*self._batch_shape, *ev) if t.shape[: t.dim() - event_dims] != self._batch_shape else t
        return list(tb.reshape(-1, *ev).unbind(0)) if self._batch_shape else [tb]

    def expand_batch(self, batch_shape):
        """The same members seen under a larger (broadcast) batch shape."""
        batch_shape = torch.Size(batch_shape)
        if batch_shape == self._batch_shape:
            return self
        idx = torch.arange(max(1, self._batch_shape.numel())).reshape(self._batch_shape if self._batch_shape else (1,))
        if not self._batch_shape:
            idx = idx.reshape(())
        idx = idx.expand(batch_shape).reshape(-1).tolist()
        return BatchLinearOperator([self.ops[i] for i in idx], batch_shape)

    # ---- protocol
    def _matmul(self, rhs):
        extra = torch.broadcast_shapes(rhs.shape[:-2], self._batch_shape)
        me = self.expand_batch(extra)
        return me._stack(o._matmul(r) for o, r in zip(me.ops, me._split(rhs, 2)))

    def matmul(self, rhs):
        if isinstance(rhs, LinearOperator):
            return MatmulLinearOperator(self, rhs)
        if rhs.dim() == 1:
            return self._matmul(rhs.unsqueeze(-1)).squeeze(-1)
        return self._matmul(rhs)

    __matmul__ = matmul

    def _transpose_nonbatch(self):
        return self._map(lambda o: o._transpose_nonbatch())

    def evaluate_kernel(self):
        return self._map(lambda o: o.evaluate_kernel())

    def detach(self):
        return self._map(lambda o: o.detach())

    def to_dense(self):
        return self._stack(o.to_dense() for o in self.ops)

    def diagonal(self, offset=0, dim1=-2, dim2=-1):
        return self._stack(o.diagonal() for o in self.ops)

    def _mul_constant(self, c):
        return BatchLinearOperator([o._mul_constant(ci.reshape(1)) for o, ci in zip(self.ops, self._split(c.reshape(*c.shape, 1) if c.dim() == len(self._batch_shape) else c, 1))],
                                   self._batch_shape)

    def mul(self, other):
        if isinstance(other, (int, float)):
            return self._map(lambda o: o.mul(other))
        other = torch.as_tensor(other)
        if other.dim() >= 2 and other.shape[-2:] == (1, 1):
            other = other.reshape(other.shape[:-2])
        bs = torch.broadcast_shapes(other.shape, self._batch_shape)
        me = self.expand_batch(bs)
        return BatchLinearOperator([o._mul_constant(ci.reshape(1)) for o, ci in zip(me.ops, other.expand(bs).reshape(-1).unbind(0))], bs)

    __mul__ = mul

    def __add__(self, other):
        if isinstance(other, ZeroLinearOperator):
            return self
        if isinstance(other, ConstantDiagLinearOperator):
            bs = torch.broadcast_shapes(other.batch_shape, self._batch_shape)
            me = self.expand_batch(bs)
            vals = other.diag_values.expand(*bs, 1).reshape(-1, 1).unbind(0) if bs else [other.diag_values]
            return BatchLinearOperator([o + ConstantDiagLinearOperator(v, other.diag_shape) for o, v in zip(me.ops, vals)], bs)
        if isinstance(other, FixedPlusConstantDiagLinearOperator):
            # fixed per-point noise + learned scalar, per member: the two parts stay apart (the scalar keeps its gradient)
            bs = torch.broadcast_shapes(other.batch_shape, self._batch_shape)
            me = self.expand_batch(bs)
            consts = me._split(other.const, 1)
            return BatchLinearOperator([o + FixedPlusConstantDiagLinearOperator(f, c) for o, f, c in zip(me.ops, me._split(other.fixed, 1), consts)], bs)
        if isinstance(other, DiagLinearOperator):
            bs = torch.broadcast_shapes(other.batch_shape, self._batch_shape)
            me = self.expand_batch(bs)
            return BatchLinearOperator([o + DiagLinearOperator(v) for o, v in zip(me.ops, me._split(other._diag, 1))], bs)
        if isinstance(other, BatchLinearOperator):
            bs = torch.broadcast_shapes(other.batch_shape, self._batch_shape)
            a, b = self.expand_batch(bs), other.expand_batch(bs)
            return BatchLinearOperator([x + y for x, y in zip(a.ops, b.ops)], bs)
        if isinstance(other, torch.Tensor):
            return BatchLinearOperator([o + v for o, v in zip(self.ops, self._split(other, 2))], self._batch_shape)
        return super().__add__(other)

    def add_jitter(self, jitter_val=1e-3):
        return self._map(lambda o: o.add_jitter(jitter_val))

    def __getitem__(self, index):
        """``op[..., rows, cols]`` slices every member; a leading integer / slice (without Ellipsis) indexes the batch."""
        if isinstance(index, tuple) and len(index) > 0 and index[0] is Ellipsis:
            inner = _strip_ellipsis(index)
            return self._map(lambda o: o[inner])
        if not isinstance(index, tuple):
            index = (index,)
        nb = len(self._batch_shape)
        if len(index) == nb + 2:
            batch_idx, inner = index[:nb], index[nb:]
        else:
            batch_idx, inner = index, None
        ids = torch.arange(len(self.ops)).reshape(self._batch_shape)[batch_idx]
        pick = [self.ops[i] for i in ids.reshape(-1).tolist()]
        pick = [o[inner] for o in pick] if inner is not None else pick
        if ids.dim() == 0:
            return pick[0]
        return BatchLinearOperator(pick, ids.shape)

    def sum(self, dim=None):
        if dim is None:
            return self._stack(o.sum() for o in self.ops).sum()
        return self._stack(o.sum(dim) for o in self.ops)

    # ---- BBMM entry points: per member, stacked
    def inv_quad_logdet(self, inv_quad_rhs=None, logdet=False, reduce_inv_quad=True):
        rs = [None] * len(self.ops) if inv_quad_rhs is None else self._split(inv_quad_rhs if inv_quad_rhs.dim() > len(self._batch_shape) + 1
                                                                              else inv_quad_rhs.unsqueeze(-1), 2)
        from .batched import batched_inv_quad_logdet, members_stackable

        if members_stackable(self.ops):
            # small members (Cholesky branch): all of them in a batch-size-independent number of launches (batched.py)
            n = self.ops[0].shape[-1]
            rr = [torch.zeros(n, 1, device=self.device, dtype=self.dtype) if r is None else r for r in rs]   # (log-det alone: a dummy column)
            iq, ld = batched_inv_quad_logdet(self.ops, rr)
            iq = iq.sum(-1) if reduce_inv_quad else iq
            iq = None if inv_quad_rhs is None else iq.reshape(*self._batch_shape, *iq.shape[1:])
            return iq, (ld.reshape(self._batch_shape) if logdet else None)
        res = [o.inv_quad_logdet(r, logdet, reduce_inv_quad) for o, r in zip(self.ops, rs)]
        iq = None if inv_quad_rhs is None else self._stack(r[0] for r in res)
        ld = self._stack(r[1] for r in res) if logdet else None
        return iq, ld

    def inv_quad(self, inv_quad_rhs, reduce_inv_quad=True):
        return self.inv_quad_logdet(inv_quad_rhs, False, reduce_inv_quad)[0]

    def logdet(self):
        return self.inv_quad_logdet(None, True)[1]

    def solve(self, rhs, lhs=None):
        squeeze = rhs.dim() == len(self._batch_shape) + 1 or rhs.dim() == 1
        r = rhs.unsqueeze(-1) if squeeze else rhs
        extra = torch.broadcast_shapes(r.shape[:-2], self._batch_shape)
        me = self.expand_batch(extra)
        sol = me._stack(o.solve(ri) for o, ri in zip(me.ops, me._split(r, 2)))
        if lhs is not None:
            sol = lhs @ sol
        return sol.squeeze(-1) if squeeze else sol

    def root_inv_decomposition(self, initial_vectors=None, test_vectors=None, method=None):
        return self._map(lambda o: o.root_inv_decomposition(method=method))

    def root_decomposition(self):
        return self._map(lambda o: o.root_decomposition())

    @property
    def root(self):
        """Stacked roots of a batch of RootLinearOperators (ranks padded with zero columns to the largest)."""
        roots = [o.root for o in self.ops]
        k = max(r.shape[-1] for r in roots)
        roots = [torch.nn.functional.pad(r, (0, k - r.shape[-1])) for r in roots]
        return self._stack(roots)

    def pivoted_cholesky(self, rank, error_tol=None, return_pivots=False):
        outs = [o.pivoted_cholesky(rank, error_tol, return_pivots) for o in self.ops]
        if return_pivots:
            return self._stack(o[0] for o in outs), self._stack(o[1] for o in outs)
        k = max(o.shape[-1] for o in outs)
        return self._stack(torch.nn.functional.pad(o, (0, k - o.shape[-1])) for o in outs)
