"""Kernel modules of the hot path: ``RBFKernel``, ``MaternKernel``, ``ScaleKernel`` (+ the ``Kernel``
base they share).  Same constructor arguments, parameter names (``raw_lengthscale``,
``raw_outputscale``), constraints and call semantics as ``gpytorch/kernels/kernel.py:84-589``,
``rbf_kernel.py``, ``matern_kernel.py``, ``scale_kernel.py`` -- but ``forward`` returns a matrix-free
:class:`~gpytorch_amd.operators.FusedKernelLinearOperator`, exactly like the in-tree precedent
``gpytorch/kernels/keops/rbf_kernel.py:44-55`` returns a ``KernelLinearOperator``.
"""
from __future__ import annotations

import math

import torch

from . import backend as B
from .functions import KernelSpec
from .module import Interval, Module, Positive
from .module import AttrGetter, AttrSetter
from .operators import BatchLinearOperator, FusedKernelLinearOperator, LinearOperator


class Kernel(Module):
    has_lengthscale = False

    def __init__(self, ard_num_dims=None, batch_shape=torch.Size([]), active_dims=None, lengthscale_prior=None,
                 lengthscale_constraint=None, eps=1e-6, **kwargs):
        super().__init__()
        self._batch_shape = torch.Size(batch_shape)
        if active_dims is not None and not torch.is_tensor(active_dims):
            active_dims = torch.tensor(active_dims, dtype=torch.long)
        self.register_buffer("active_dims", active_dims)
        self.ard_num_dims = ard_num_dims
        self.eps = eps
        if self.has_lengthscale:
            n_ls = 1 if ard_num_dims is None else ard_num_dims
            self.register_parameter("raw_lengthscale", torch.nn.Parameter(torch.zeros(*self._batch_shape, 1, n_ls)))
            self.register_constraint("raw_lengthscale", Positive() if lengthscale_constraint is None else lengthscale_constraint)
            if lengthscale_prior is not None:
                self.register_prior("lengthscale_prior", lengthscale_prior, AttrGetter("lengthscale"), AttrSetter("_set_lengthscale"))

    @property
    def batch_shape(self):
        return self._batch_shape

    @property
    def dtype(self):
        """kernel.py:276-285."""
        if self.has_lengthscale:
            return self.lengthscale.dtype
        dtypes = {p.dtype for p in self.parameters()}
        if len(dtypes) > 1:
            raise RuntimeError(f"The kernel's parameters have multiple dtypes: {dtypes}.")
        return dtypes.pop() if dtypes else torch.get_default_dtype()

    @property
    def device(self):
        devices = {p.device for p in self.parameters()}
        if len(devices) > 1:
            raise RuntimeError(f"The kernel's parameters are on multiple devices: {devices}.")
        return devices.pop() if devices else torch.device("cpu")

    def named_sub_kernels(self):
        """Directly held member kernels (``kernel.py:405-414``; the list-holding compositions override ``__getitem__`` / ``expand_batch`` themselves)."""
        for name, module in self.named_children():
            if isinstance(module, Kernel):
                yield name, module

    def _batch_parameters(self):
        """Own parameters and those buffers that carry the batch shape (``active_dims`` is a buffer WITHOUT it)."""
        yield from self.named_parameters(recurse=False)
        nb = len(self._batch_shape)
        for name, buf in self.named_buffers(recurse=False):
            if buf is not None and name != "active_dims" and buf.dim() > nb and buf.shape[:nb] == self._batch_shape:
                yield name, buf

    def __getitem__(self, index):
        """``kernel.py:556-590``: the kernel of the indexed batch members -- every batch-shaped parameter indexed, the batch shape shortened by the
        dimensions the index removed, member kernels indexed the same way."""
        if len(self._batch_shape) == 0:
            return self
        import copy

        new = copy.deepcopy(self)
        index = index if isinstance(index, tuple) else (index,)
        for name, old in self._batch_parameters():
            t = getattr(new, name)
            t.data = t.data[index]
            new._batch_shape = t.shape[: len(self._batch_shape) - (old.dim() - t.dim())]
        for name, sub in self.named_sub_kernels():
            setattr(new, name, sub[index])
        return new

    def expand_batch(self, *sizes):
        """``kernel.py:354-403``: the same kernel with its parameters expanded to a larger batch shape."""
        if len(sizes) == 1 and hasattr(sizes[0], "__iter__"):
            new_shape = torch.Size(sizes[0])
        elif all(isinstance(v, int) for v in sizes):
            new_shape = torch.Size(sizes)
        else:
            raise RuntimeError(f"Invalid arguments {sizes} to expand_batch.")
        if new_shape == self._batch_shape:
            return self
        try:
            torch.broadcast_shapes(new_shape, self._batch_shape)
        except RuntimeError:
            raise RuntimeError(f"Cannot expand a kernel with batch shape {self._batch_shape} to new shape {new_shape}")
        import copy

        new = copy.deepcopy(self)
        nb = len(self._batch_shape)
        for name, old in self._batch_parameters():
            getattr(new, name).data = old.data.expand(*new_shape, *old.shape[nb:]).clone()
        new._batch_shape = new_shape
        for name, sub in self.named_sub_kernels():
            setattr(new, name, sub.expand_batch(new_shape))
        return new

    @property
    def lengthscale(self):
        return self._get_transformed("raw_lengthscale") if self.has_lengthscale else None

    @lengthscale.setter
    def lengthscale(self, value):
        self._set_lengthscale(value)

    def _set_lengthscale(self, value):
        if not self.has_lengthscale:
            raise RuntimeError("Kernel has no lengthscale.")
        self._set_transformed("raw_lengthscale", value)

    @property
    def is_stationary(self):
        return self.has_lengthscale

    def forward(self, x1, x2, diag=False, **params):
        raise NotImplementedError

    def __call__(self, x1, x2=None, diag=False, last_dim_is_batch=False, **params):
        """``kernel.py:454-534``: select active dims, promote 1-D inputs to [n, 1], default x2 = x1; leading dimensions of
        the inputs (and the kernel's ``batch_shape``) are batch dimensions (``kernel.py:163-208``)."""
        x1_, x2_ = x1, (None if x2 is x1 else x2)
        in_forward = last_dim_is_batch and getattr(self, "dims_as_batch_in_forward", False)
        if last_dim_is_batch and not in_forward:  # kernel.py:336-338: every input dimension becomes its own batch member, [..., d, n, 1]
            x1_ = x1_.transpose(-1, -2).unsqueeze(-1)
            x2_ = None if x2_ is None else x2_.transpose(-1, -2).unsqueeze(-1)
        if x1_.dim() == 1:
            x1_ = x1_.unsqueeze(1)
        if x2_ is not None and x2_.dim() == 1:
            x2_ = x2_.unsqueeze(1)
        if self.active_dims is not None:
            x1_ = x1_.index_select(-1, self.active_dims)
            if x2_ is not None:
                x2_ = x2_.index_select(-1, self.active_dims)
        if x2_ is None:
            x2_ = x1_
        elif x1_.shape[-1] != x2_.shape[-1]:
            raise RuntimeError("x1_ and x2_ must have the same number of dimensions!")
        if self.ard_num_dims is not None and self.ard_num_dims != x1_.shape[-1]:
            raise RuntimeError(f"Expected the input to have {self.ard_num_dims} dimensionality (based on ard_num_dims). Got {x1_.shape[-1]}.")
        if in_forward:      # (the stationary families scale by their per-dimension lengthscales first, as the reference's forward does)
            params["last_dim_is_batch"] = True
        return self.forward(x1_, x2_, diag=diag, **params)

    @property
    def prediction_strategy(self):
        from .models import DefaultPredictionStrategy

        return DefaultPredictionStrategy

    # ---- composition (kernels/kernel.py:563-589: ``+`` -> AdditiveKernel, ``*`` -> ProductKernel)
    def __add__(self, other):
        kernels = list(self.kernels) if isinstance(self, AdditiveKernel) else [self]
        kernels += list(other.kernels) if isinstance(other, AdditiveKernel) else [other]
        return AdditiveKernel(*kernels)

    def __mul__(self, other):
        kernels = list(self.kernels) if isinstance(self, ProductKernel) else [self]
        kernels += list(other.kernels) if isinstance(other, ProductKernel) else [other]
        return ProductKernel(*kernels)

    def _select(self, x):
        """Promote 1-D inputs and apply ``active_dims`` (the input handling of ``__call__``)."""
        x = x.unsqueeze(1) if x.dim() == 1 else x
        return x if self.active_dims is None else x.index_select(-1, self.active_dims)

    def rbf_features(self, x):
        """phi(x) with  k(x, x') = exp(-1/2 |phi(x) - phi(x')|^2)  for kernels of the squared-exponential family (RBF: x / l;
        Periodic: (cos, sin)(2 pi x / p) / sqrt(l); their products: concatenation), or None.  It lets compositions run on the
        fused RBF kernels: hyper-parameters reach the objective through phi, whose gradient is the fused input gradient."""
        return None


class _StationaryFused(Kernel):
    has_lengthscale = True
    kind = None

    def _shift(self, x1):
        # stationary kernel: any common shift is exact; centring keeps |z| small, which the Gram-form
        # generation kernel needs (the reference's sq_dist centres for the same reason, kernel.py:29-30)
        return x1.detach().mean(dim=-2)

    def _make_spec(self, x1, b=None, batch=None):
        """Non-tensor description of the operator for batch member ``b`` (families with a shape parameter add it here)."""
        return KernelSpec(self.kind, self._shift(x1))

    dims_as_batch_in_forward = True

    def forward(self, x1, x2, diag=False, last_dim_is_batch=False, **params):
        # float32 with d <= 16: fused MFMA / VALU kernels; float64 or d > 16: generic path (backend.kv_chunked)
        ls = self.lengthscale
        same = x2 is x1
        nd = None
        if last_dim_is_batch:
            # rbf_kernel.py:78-81 + kernel.py:336-338 (deprecated in the reference, kept): the inputs are divided by the lengthscale of THEIR
            # dimension, then every dimension becomes a batch member [..., d, n, 1] -- so the lengthscales move to the batch with them
            nd = x1.shape[-1]
            x1 = x1.transpose(-1, -2).unsqueeze(-1)
            x2 = x1 if same else x2.transpose(-1, -2).unsqueeze(-1)
            ls = ls.expand(*ls.shape[:-1], nd).transpose(-1, -2).unsqueeze(-1)
        batch = torch.broadcast_shapes(x1.shape[:-2], x2.shape[:-2], ls.shape[:-2])
        if not batch:
            op = FusedKernelLinearOperator(x1, x2, self._make_spec(x1), ls)
            return op.diagonal() if diag else op
        # batch mode: one fused operator per batch member (inputs and lengthscales broadcast against each other)
        x1b = x1.expand(*batch, *x1.shape[-2:]).reshape(-1, *x1.shape[-2:])
        x2b = x1b if same else x2.expand(*batch, *x2.shape[-2:]).reshape(-1, *x2.shape[-2:])
        lsb = ls.expand(*batch, *ls.shape[-2:]).reshape(-1, *ls.shape[-2:])
        ops = []
        for b in range(x1b.shape[0]):
            xa = x1b[b]
            xb = xa if same else x2b[b]
            if nd is None:
                spec = self._make_spec(xa, b, batch)
            else:               # (shape parameters belong to the kernel's own batch: the dimension index is the last batch dimension)
                spec = self._make_spec(xa, b // nd, batch[:-1]) if len(batch) > 1 else self._make_spec(xa)
            ops.append(FusedKernelLinearOperator(xa, xb, spec, lsb[b]))
        op = BatchLinearOperator(ops, batch)
        return op.diagonal() if diag else op


class RBFKernel(_StationaryFused):
    r"""k(x, x') = exp(-1/2 (x - x')^T Theta^-2 (x - x'))  (``gpytorch/kernels/rbf_kernel.py:14-85``)."""

    kind = "rbf"

    def rbf_features(self, x):
        return self._select(x) / self.lengthscale


def _unit_lengthscale(x):
    return torch.ones(1, 1, device=x.device, dtype=x.dtype)


def _feature_operator(f1, f2, same):
    """exp(-1/2 |f1_i - f2_j|^2) as a fused RBF operator with unit lengthscale over feature clouds (batch-aware)."""
    if f1.dim() == 2:
        f2 = f1 if same else f2
        return FusedKernelLinearOperator(f1, f2, KernelSpec("rbf", f1.detach().mean(dim=-2)), _unit_lengthscale(f1))
    batch = torch.broadcast_shapes(f1.shape[:-2], f2.shape[:-2])
    f1b = f1.expand(*batch, *f1.shape[-2:]).reshape(-1, *f1.shape[-2:])
    f2b = f1b if same else f2.expand(*batch, *f2.shape[-2:]).reshape(-1, *f2.shape[-2:])
    ops = [_feature_operator(f1b[b], f1b[b] if same else f2b[b], same) for b in range(f1b.shape[0])]
    return BatchLinearOperator(ops, batch)


def _feature_diag(f1, f2):
    """k(x1_i, x2_i) = exp(-1/2 |phi(x1_i) - phi(x2_i)|^2): the ``diag=True`` value for two DIFFERENT equally long inputs (kernels/kernel.py:318-330
    evaluates the elementwise diagonal, not the all-ones diagonal of K(x, x))."""
    return (f1 - f2).pow(2).sum(-1).mul(-0.5).exp()


def _cat_features(fs):
    """Feature maps of the members of a product side by side; members with and without a batch shape broadcast against each other."""
    batch = torch.broadcast_shapes(*[f.shape[:-1] for f in fs])
    return torch.cat([f.expand(*batch, f.shape[-1]) for f in fs], dim=-1)


class PeriodicKernel(Kernel):
    r"""k(x, x') = exp(-2 sum_q sin^2(pi (x_q - x'_q) / p_q) / l_q)   (``gpytorch/kernels/periodic_kernel.py:14-142``, the
    KeOps twin ``kernels/keops/periodic_kernel.py``).  Since sin^2(a - b) = (1 - cos 2(a - b)) / 2, the exponent is
    -1/2 |phi(x) - phi(x')|^2 with phi_q = (cos, sin)(2 pi x_q / p_q) / sqrt(l_q): a fused RBF operator over 2 d features
    (d <= 8 on the fused float32 kernels)."""

    has_lengthscale = True

    def __init__(self, period_length_prior=None, period_length_constraint=None, **kwargs):
        super().__init__(**kwargs)
        n_p = 1 if self.ard_num_dims is None else self.ard_num_dims
        self.register_parameter("raw_period_length", torch.nn.Parameter(torch.zeros(*self._batch_shape, 1, n_p)))
        self.register_constraint("raw_period_length", Positive() if period_length_constraint is None else period_length_constraint)
        if period_length_prior is not None:
            self.register_prior("period_length_prior", period_length_prior, AttrGetter("period_length"), AttrSetter("_set_period_length"))

    @property
    def period_length(self):
        return self._get_transformed("raw_period_length")

    @period_length.setter
    def period_length(self, value):
        self._set_period_length(value)

    def _set_period_length(self, value):
        self._set_transformed("raw_period_length", value)

    def rbf_features(self, x):
        x = self._select(x)
        a = x * (2.0 * math.pi / self.period_length)
        return torch.cat([a.cos(), a.sin()], dim=-1) / torch.cat([self.lengthscale.sqrt().expand_as(a[..., :1, :])] * 2, dim=-1)

    def forward(self, x1, x2, diag=False, **params):
        same = x2 is x1
        if diag:
            if same or (x1.shape == x2.shape and torch.equal(x1, x2)):
                return torch.ones(x1.shape[:-1], device=x1.device, dtype=x1.dtype)
            return _feature_diag(self.rbf_features(x1), self.rbf_features(x2))
        f1 = self.rbf_features(x1)
        return _feature_operator(f1, f1 if same else self.rbf_features(x2), same)

    def __call__(self, x1, x2=None, diag=False, **params):
        # active_dims are applied inside rbf_features (so that products can concatenate members with different active_dims)
        x2 = x1 if x2 is None else x2
        return self.forward(x1, x2, diag=diag, **params)


class MaternKernel(_StationaryFused):
    r"""Matern nu in {1/2, 3/2, 5/2} (``gpytorch/kernels/matern_kernel.py:14-110``); inputs are centred by
    the mean of x1 first, as the reference does (``matern_kernel.py:94-97``)."""

    def __init__(self, nu: float = 2.5, **kwargs):
        if nu not in {0.5, 1.5, 2.5}:
            raise RuntimeError("nu expected to be 0.5, 1.5, or 2.5")
        super().__init__(**kwargs)
        self.nu = nu

    @property
    def kind(self):
        return B.NU_TO_KIND[self.nu]

    def _shift(self, x1):
        return x1.detach().mean(dim=-2)


class RQKernel(_StationaryFused):
    r"""k(x, x') = (1 + (x - x')^T Theta^-2 (x - x') / (2 alpha))^-alpha   (``gpytorch/kernels/rq_kernel.py:14-86``).  A native
    covariance family of the fused float32 kernels (``KIND_RQ``: one ``v_log_f32`` + one ``v_exp_f32`` per pair) and of the float64 / d > 16
    generic path; alpha is a learnable shape parameter whose gradient comes out of the same derivative pass as the lengthscales'."""

    kind = "rq"

    def __init__(self, alpha_constraint=None, **kwargs):
        super().__init__(**kwargs)
        self.register_parameter("raw_alpha", torch.nn.Parameter(torch.zeros(*self._batch_shape, 1)))
        self.register_constraint("raw_alpha", Positive() if alpha_constraint is None else alpha_constraint)

    @property
    def alpha(self):
        return self._get_transformed("raw_alpha")

    @alpha.setter
    def alpha(self, value):
        self._set_transformed("raw_alpha", value)

    def _make_spec(self, x1, b=None, batch=None):
        a = self.alpha
        if b is not None:
            a = a.expand(*batch, 1).reshape(-1, 1)[b]
        return KernelSpec("rq", self._shift(x1), param=a)


class ScaleKernel(Kernel):
    r"""K_scaled = outputscale * K_orig  (``gpytorch/kernels/scale_kernel.py:20-124``)."""

    def __init__(self, base_kernel, outputscale_prior=None, outputscale_constraint=None, **kwargs):
        if base_kernel.active_dims is not None:
            kwargs["active_dims"] = base_kernel.active_dims
        super().__init__(**kwargs)
        self.base_kernel = base_kernel
        self.register_parameter("raw_outputscale", torch.nn.Parameter(torch.zeros(self._batch_shape)))
        self.register_constraint("raw_outputscale", Positive() if outputscale_constraint is None else outputscale_constraint)
        if outputscale_prior is not None:
            self.register_prior("outputscale_prior", outputscale_prior, AttrGetter("outputscale"), AttrSetter("_set_outputscale"))

    @property
    def is_stationary(self):
        return self.base_kernel.is_stationary

    @property
    def outputscale(self):
        return self._get_transformed("raw_outputscale")

    @outputscale.setter
    def outputscale(self, value):
        self._set_outputscale(value)

    def _set_outputscale(self, value):
        self._set_transformed("raw_outputscale", value)

    def __call__(self, x1, x2=None, diag=False, **params):
        # active_dims were inherited from the base kernel: let the base kernel apply them once
        return self.forward(x1, x2, diag=diag, **params)

    def forward(self, x1, x2, diag=False, **params):
        orig = self.base_kernel(x1, x2, diag=diag, **params)
        os_ = self.outputscale
        if params.get("last_dim_is_batch", False):      # scale_kernel.py:110-112: the input dimensions became the LAST batch dimension
            os_ = os_.unsqueeze(-1)
        if diag:
            return orig * os_.unsqueeze(-1)
        if isinstance(orig, LinearOperator):  # scale_kernel.py:117-118: outputscale.view(*batch, 1, 1)
            return orig.mul(os_.reshape(1) if os_.numel() == 1 and not orig.batch_shape else os_)
        return orig * os_.reshape(*os_.shape, 1, 1)

    @property
    def prediction_strategy(self):
        return self.base_kernel.prediction_strategy


class MultiDeviceKernel(Kernel):
    r"""Constructor-compatible stand-in for ``gpytorch.kernels.MultiDeviceKernel`` (``kernels/multi_device_kernel.py:14-92``).

    The reference scatters row chunks of x1 over ``device_ids`` inside ONE process (``DataParallel``: module replicas re-created on
    every forward, peer copies of V on every product) and concatenates dense chunks.  Here multi-GPU means one process per GPU
    (``torchrun``; RCCL): the wrapped kernel returns the same fused operator as without the wrapper, and the operator shards its
    work at solve time -- probe columns of the MLL over ``settings.sharding.probe_group``, the few-column posterior solves by rows
    over ``settings.sharding.row_group`` (``distributed.py``).  Constructing this kernel in a process group of more than one rank
    installs the automatic layout policy over WORLD (``settings.sharding("auto")``, unless a sharding scope is already set): the closest
    equivalent of "allocate the covariance on these devices" -- the user names devices, never a probe-share x row-block grid; in a single
    process with several ``device_ids`` it warns once and runs on the inputs' device.
    The wrapped kernel is registered as ``module`` (the name ``DataParallel`` uses), so state-dict keys match the reference's."""

    def __init__(self, base_kernel, device_ids, output_device=None, create_cuda_context=True, **kwargs):
        super().__init__(**kwargs)
        self.module = base_kernel
        self.device_ids = list(device_ids)
        self.output_device = output_device if output_device is not None else (self.device_ids[0] if self.device_ids else None)
        import torch.distributed as dist

        from . import settings

        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # process-global, like every setting here (and like the reference's settings): said out loud once, undone by `release()`.  The user of
            # the reference's MultiDeviceKernel names devices, not a layout: the "auto" policy of settings.sharding takes that role -- every MLL
            # evaluation picks its probe-share x row-block grid from (n, probes) (distributed.choose_grid), the posterior solves are row-sharded over WORLD
            import warnings

            self._installed = []
            if not settings.sharding._auto and settings.sharding._probe_group is None and settings.sharding._row_group is None \
                    and settings.sharding._mll_row_group is None:
                settings.sharding._auto = True
                self._installed = ["auto"]
                warnings.warn("gpytorch_amd.kernels.MultiDeviceKernel installed the WORLD process group with the automatic layout policy "
                              "(settings.sharding('auto')) for EVERY model of this process; MultiDeviceKernel.release() (or a settings.sharding(...) "
                              "scope) undoes it.", RuntimeWarning)
        elif len(self.device_ids) > 1:
            import warnings

            warnings.warn("gpytorch_amd.kernels.MultiDeviceKernel: multi-GPU runs are one process per GPU (torchrun + settings.sharding); "
                          "this single process evaluates the kernel on the device of its inputs.", RuntimeWarning)

    def release(self):
        """Take back the process groups this constructor installed in ``settings.sharding`` (no-op if it installed none)."""
        from . import settings

        if getattr(self, "_installed", []):
            settings.sharding._auto = False
        self._installed = []

    @property
    def base_kernel(self):
        return self.module

    @property
    def is_stationary(self):
        return self.module.is_stationary

    def __call__(self, x1, x2=None, diag=False, **params):
        return self.forward(x1, x2, diag=diag, **params)     # (the wrapped kernel applies its own active_dims)

    def forward(self, x1, x2, diag=False, **params):
        return self.module(x1, x2, diag=diag, **params)

    def num_outputs_per_input(self, x1, x2):
        f = getattr(self.module, "num_outputs_per_input", None)
        return 1 if f is None else f(x1, x2)

    @property
    def prediction_strategy(self):
        return self.module.prediction_strategy


def _index_members(kernel, index):
    """``kernel.py:626-631`` / ``:684-688``: a sum / product of kernels indexes every member."""
    import copy

    new = copy.deepcopy(kernel)
    for i, member in enumerate(kernel.kernels):
        new.kernels[i] = member[index]
    return new


class AdditiveKernel(Kernel):
    """K = sum_i K_i (``kernels/kernel.py:592-632``): the members stay matrix-free; their sum is a
    :class:`~gpytorch_amd.operators.SumFusedLinearOperator` whose products, solves and log-determinants add the members'
    fused products."""

    def __init__(self, *kernels):
        super().__init__()
        self.kernels = torch.nn.ModuleList(kernels)

    @property
    def is_stationary(self):
        return all(k.is_stationary for k in self.kernels)

    def __getitem__(self, index):
        return _index_members(self, index)

    def __call__(self, x1, x2=None, diag=False, **params):
        return self.forward(x1, x1 if x2 is None else x2, diag=diag, **params)

    def forward(self, x1, x2, diag=False, **params):
        from .composite import SumFusedLinearOperator

        terms = [k(x1, x2, diag=diag, **params) for k in self.kernels]
        if diag:
            return sum(terms[1:], terms[0])
        return SumFusedLinearOperator.of(terms)


def _se_family(kernel):
    """(feature function, outputscale or None) if ``kernel`` is a product of squared-exponential-family members, else None."""
    if isinstance(kernel, ScaleKernel):
        inner = _se_family(kernel.base_kernel)
        if inner is None:
            return None
        return inner[0], (kernel.outputscale if inner[1] is None else kernel.outputscale * inner[1])
    if isinstance(kernel, ProductKernel):
        parts = [_se_family(k) for k in kernel.kernels]
        if any(p is None for p in parts):
            return None
        scale = None
        for _, sc in parts:
            if sc is not None:
                scale = sc if scale is None else scale * sc
        return (lambda x, parts=parts: _cat_features([f(x) for f, _ in parts])), scale
    if kernel.rbf_features.__func__ is not Kernel.rbf_features:
        return kernel.rbf_features, None
    return None


class ProductKernel(Kernel):
    """K = prod_i K_i elementwise (``kernels/kernel.py:634-688``).  Products of squared-exponential-family members (RBF,
    Periodic, their ScaleKernels) are ONE fused RBF operator over the concatenated feature maps -- exact, matrix-free, at most 16
    feature dimensions.  Any other product is formed densely (the reference, too, densifies whenever x1 != x2) and is meant
    for small problems."""

    def __init__(self, *kernels):
        super().__init__()
        self.kernels = torch.nn.ModuleList(kernels)

    @property
    def is_stationary(self):
        return all(k.is_stationary for k in self.kernels)

    def __getitem__(self, index):
        return _index_members(self, index)

    def __call__(self, x1, x2=None, diag=False, **params):
        return self.forward(x1, x1 if x2 is None else x2, diag=diag, **params)

    def forward(self, x1, x2, diag=False, **params):
        fam = _se_family(self)
        if fam is not None:
            feat, scale = fam
            same = x2 is x1
            if diag:
                if same or (x1.shape == x2.shape and torch.equal(x1, x2)):
                    one = torch.ones(x1.shape[:-1] if x1.dim() > 1 else x1.shape, device=x1.device, dtype=x1.dtype)
                else:     # two different inputs: the elementwise diagonal (test/kernels/test_additive_and_product_kernels.py:127-157)
                    one = _feature_diag(feat(x1), feat(x2))
                return one if scale is None else one * scale.unsqueeze(-1)
            f1 = feat(x1)
            if f1.shape[-1] <= B.MAX_INPUT_DIM or f1.dtype == torch.float64:
                op = _feature_operator(f1, f1 if same else feat(x2), same)
                return op if scale is None else op.mul(scale.reshape(1) if scale.numel() == 1 and not op.batch_shape else scale)
        from .operators import DenseLinearOperator, to_dense

        res = None
        for k in self.kernels:
            term = k(x1, x2, diag=diag, **params)
            term = term if diag else to_dense(term)
            res = term if res is None else res * term
        return res if diag else DenseLinearOperator(res)


__all__ = ["Kernel", "RBFKernel", "MaternKernel", "RQKernel", "PeriodicKernel", "ScaleKernel", "AdditiveKernel", "ProductKernel"]
_ = (math, Interval)
