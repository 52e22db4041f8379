"""Host-side glue mirroring ``gpytorch/module.py`` and ``gpytorch/constraints/constraints.py`` just far
enough for the ExactGP hot path: raw parameters + constraints (softplus / sigmoid transforms),
``initialize(**kwargs)``, optional priors, ``named_priors`` / ``added_loss_terms`` iteration used by
``ExactMarginalLogLikelihood._add_other_terms`` (``gpytorch/mlls/exact_marginal_log_likelihood.py:41-52``).
Pure PyTorch; nothing here is on the hot path.
"""
from __future__ import annotations

import math

import torch
from torch import nn


def inv_softplus(x: torch.Tensor) -> torch.Tensor:
    """``gpytorch/utils/transforms.py:8-9``."""
    return x + torch.log(-torch.expm1(-x))


def _sigmoid(x):
    return torch.sigmoid(x)


def _inv_sigmoid(x):
    return torch.log(x) - torch.log1p(-x)


def _softplus(x):
    return torch.nn.functional.softplus(x)


_KNOWN_INVERSES = {torch.exp: torch.log, torch.sigmoid: _inv_sigmoid, _sigmoid: _inv_sigmoid, _softplus: None, torch.nn.functional.softplus: None}


class Interval(nn.Module):
    """``gpytorch/constraints/constraints.py:17-150``: a parameter kept inside (lower, upper) by ``transform`` of its raw value -- a map onto (0, 1)
    when both bounds are finite (sigmoid by default), onto (0, inf) when one is (softplus by default); ``transform=None`` registers the bounds
    without enforcing them; ``initial_value`` (a constrained value) is stored as the RAW value the parameter starts at."""

    def __init__(self, lower_bound, upper_bound, transform=_sigmoid, inv_transform=_inv_sigmoid, initial_value=None):
        dtype = torch.get_default_dtype()
        lower_bound = torch.as_tensor(lower_bound).to(dtype)
        upper_bound = torch.as_tensor(upper_bound).to(dtype)
        if torch.any(torch.ge(lower_bound, upper_bound)):
            raise ValueError("Got parameter bounds with empty intervals.")
        if type(self) is Interval and (torch.max(upper_bound) == math.inf or torch.min(lower_bound) == -math.inf):
            raise ValueError("Cannot make an Interval directly with non-finite bounds. Use a derived class like GreaterThan or LessThan instead.")
        super().__init__()
        self.register_buffer("lower_bound", lower_bound)
        self.register_buffer("upper_bound", upper_bound)
        self._transform = transform
        self._inv_transform = inv_transform
        if transform is not None and inv_transform is None:
            known = _KNOWN_INVERSES.get(transform)
            if known is None and transform in (_softplus, torch.nn.functional.softplus):
                known = inv_softplus
            if known is None:
                raise RuntimeError("Must specify inv_transform for custom transforms")
            self._inv_transform = known
        self._initial_value = None if initial_value is None else self.inverse_transform(torch.as_tensor(initial_value))

    @property
    def enforced(self) -> bool:
        return self._transform is not None

    def check(self, tensor) -> bool:
        return bool(torch.all(tensor <= self.upper_bound) and torch.all(tensor >= self.lower_bound))

    def check_raw(self, tensor) -> bool:
        return self.check(self.transform(tensor))

    def intersect(self, other):
        """The intersection of two constraints with the same transform (``constraints.py:87-101``)."""
        if self._transform != other._transform:
            raise RuntimeError("Cant intersect Interval constraints with conflicting transforms!")
        return Interval(torch.max(self.lower_bound, other.lower_bound), torch.min(self.upper_bound, other.upper_bound))

    def transform(self, tensor):
        if not self.enforced:
            return tensor
        return self._transform(tensor) * (self.upper_bound - self.lower_bound) + self.lower_bound

    def inverse_transform(self, tensor):
        if not self.enforced:
            return tensor
        return self._inv_transform((tensor - self.lower_bound) / (self.upper_bound - self.lower_bound))

    @property
    def initial_value(self):
        """The RAW value the parameter starts at (None if no initial value was given)."""
        return self._initial_value

    def __iter__(self):
        yield self.lower_bound
        yield self.upper_bound


class GreaterThan(Interval):
    """``constraints.py:153-178``: transform(raw) + lower_bound, softplus by default."""

    def __init__(self, lower_bound, transform=_softplus, inv_transform=inv_softplus, initial_value=None):
        super().__init__(lower_bound, math.inf, transform=transform, inv_transform=inv_transform, initial_value=initial_value)

    def transform(self, tensor):
        return self._transform(tensor) + self.lower_bound if self.enforced else tensor

    def inverse_transform(self, tensor):
        return self._inv_transform(tensor - self.lower_bound) if self.enforced else tensor


class Positive(GreaterThan):
    """``constraints.py:181-194``."""

    def __init__(self, transform=_softplus, inv_transform=inv_softplus, initial_value=None):
        super().__init__(0.0, transform=transform, inv_transform=inv_transform, initial_value=initial_value)

    def transform(self, tensor):
        return self._transform(tensor) if self.enforced else tensor

    def inverse_transform(self, tensor):
        return self._inv_transform(tensor) if self.enforced else tensor


class LessThan(Interval):
    """``constraints.py:197-222``: upper_bound - transform(-raw)."""

    def __init__(self, upper_bound, transform=_softplus, inv_transform=inv_softplus, initial_value=None):
        super().__init__(-math.inf, upper_bound, transform=transform, inv_transform=inv_transform, initial_value=initial_value)

    def transform(self, tensor):
        return -self._transform(-tensor) + self.upper_bound if self.enforced else tensor

    def inverse_transform(self, tensor):
        return -self._inv_transform(-(tensor - self.upper_bound)) if self.enforced else tensor


class AttrGetter:
    """``closure(module) -> module.<name>``: what a prior is evaluated on.  A class, not a lambda, so that modules with priors pickle
    (``test/kernels/test_scale_kernel.py:137-141``, ``test/likelihoods/test_gaussian_likelihood.py:23-27``: pickle round trips with a prior)."""

    def __init__(self, name):
        self.name = name

    def __call__(self, module):
        return getattr(module, self.name)


class AttrSetter:
    """``setting_closure(module, value) -> module.<method>(*args, value)``: how a value sampled from a prior is written back."""

    def __init__(self, method, *args):
        self.method, self.args = method, args

    def __call__(self, module, value):
        return getattr(module, self.method)(*self.args, value)


class Module(nn.Module):
    """Parameter / constraint / prior registry (``gpytorch/module.py``)."""

    def __init__(self):
        super().__init__()
        self._constraints_map = {}
        self._priors_reg = {}
        self._added_loss_terms = {}

    def __call__(self, *inputs, **kwargs):  # module.py:82-86
        outputs = self.forward(*inputs, **kwargs)
        if isinstance(outputs, list):
            return [o for o in outputs]
        return outputs

    def register_constraint(self, param_name: str, constraint: Interval):
        self.add_module(param_name + "_constraint", constraint)
        self._constraints_map[param_name] = constraint
        if constraint.initial_value is not None:       # (already a raw value: module.py:348-349)
            self.initialize(**{param_name: constraint.initial_value})

    def constraint_for(self, param_name: str):
        return self._constraints_map.get(param_name)

    def constraint_for_parameter_name(self, param_name: str):
        """module.py:206-227: the constraint of a (possibly dotted, nested) parameter name, or None."""
        base, _, name = param_name.rpartition(".")
        mod = self
        for part in (base.split(".") if base else []):
            mod = getattr(mod, part, None)
            if mod is None:
                return None
        return mod.constraint_for(name) if isinstance(mod, Module) else None

    def named_parameters_and_constraints(self):
        """module.py:229-231."""
        for name, param in self.named_parameters():
            yield name, param, self.constraint_for_parameter_name(name)

    def named_constraints(self):
        for name, mod in self.named_modules():
            if isinstance(mod, Module):
                for pname, c in mod._constraints_map.items():
                    yield (f"{name}.{pname}_constraint" if name else f"{pname}_constraint"), c

    def register_prior(self, name, prior, param_or_closure, setting_closure=None):
        from .priors import Prior

        if not isinstance(prior, Prior):     # (kernels/scale_kernel.py:91-92, periodic_kernel.py:107-108, index_kernel.py:74-75: the reference's type check)
            raise TypeError("Expected gpytorch.priors.Prior but got " + type(prior).__name__)
        closure = AttrGetter(param_or_closure) if isinstance(param_or_closure, str) else param_or_closure
        self.add_module(name, prior) if isinstance(prior, nn.Module) else None
        self._priors_reg[name] = (prior, closure, setting_closure)

    def named_priors(self, memo=None, prefix=""):
        """Yields (name, module, prior, closure, setting_closure) like ``gpytorch.Module.named_priors``."""
        memo = set() if memo is None else memo
        for name, (prior, closure, sc) in self._priors_reg.items():
            if prior not in memo:
                memo.add(prior)
                yield prefix + ("." if prefix else "") + name, self, prior, closure, sc
        for mname, module in self.named_children():
            if isinstance(module, Module):
                yield from module.named_priors(memo, prefix + ("." if prefix else "") + mname)

    def added_loss_terms(self):
        for t in self._added_loss_terms.values():
            yield t
        for module in self.children():
            if isinstance(module, Module):
                yield from module.added_loss_terms()

    def initialize(self, **kwargs):
        """``module.py:122-191``: set (raw or transformed) parameter values by name."""
        for name, val in kwargs.items():
            if "." in name:
                child, rest = name.split(".", 1)
                getattr(self, child).initialize(**{rest: val})
                continue
            if name not in self._parameters and name not in self._buffers:
                setter = getattr(type(self), name, None)
                if isinstance(setter, property) and setter.fset is not None:
                    setter.fset(self, val)
                    continue
                raise AttributeError(f"Unknown parameter {name} for {self.__class__.__name__}")
            param = getattr(self, name)
            val_t = torch.as_tensor(val, dtype=param.dtype, device=param.device)
            with torch.no_grad():
                try:
                    param.copy_(val_t.expand_as(param))
                except RuntimeError:
                    # module.py:170-178: a value that does not broadcast but has the parameter's element count is re-viewed (e.g. a [1, 1, 1]
                    # lengthscale given to a kernel without a batch shape, test/kernels/test_periodic_kernel.py:56-62)
                    if val_t.numel() != param.numel():
                        raise
                    param.copy_(val_t.reshape(param.shape))
        return self

    def _set_transformed(self, raw_name: str, value):
        param = getattr(self, raw_name)
        value = torch.as_tensor(value, dtype=param.dtype, device=param.device)
        c = self.constraint_for(raw_name)
        raw = c.inverse_transform(value) if c is not None else value
        self.initialize(**{raw_name: raw})

    def _get_transformed(self, raw_name: str):
        param = getattr(self, raw_name)
        c = self.constraint_for(raw_name)
        return c.transform(param) if c is not None else param

    def hyperparameters(self):
        yield from self.parameters()
