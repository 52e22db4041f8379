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


class Interval(nn.Module):
    """``gpytorch/constraints/constraints.py``: sigmoid-transformed box constraint."""

    def __init__(self, lower_bound, upper_bound, initial_value=None):
        super().__init__()
        self.register_buffer("lower_bound", torch.as_tensor(float(lower_bound)))
        self.register_buffer("upper_bound", torch.as_tensor(float(upper_bound)))
        self._initial_value = initial_value

    @property
    def enforced(self):
        return True

    def transform(self, tensor):
        return torch.sigmoid(tensor) * (self.upper_bound - self.lower_bound) + self.lower_bound

    def inverse_transform(self, tensor):
        p = (tensor - self.lower_bound) / (self.upper_bound - self.lower_bound)
        return torch.log(p) - torch.log1p(-p)

    @property
    def initial_value(self):
        return self._initial_value


class GreaterThan(Interval):
    """``constraints.py:160-178``: softplus(raw) + lower_bound."""

    def __init__(self, lower_bound, initial_value=None):
        super().__init__(lower_bound, math.inf, initial_value)

    def transform(self, tensor):
        return torch.nn.functional.softplus(tensor) + self.lower_bound

    def inverse_transform(self, tensor):
        return inv_softplus(tensor - self.lower_bound)


class Positive(GreaterThan):
    """``constraints.py:181-194``."""

    def __init__(self, initial_value=None):
        super().__init__(0.0, initial_value)

    def transform(self, tensor):
        return torch.nn.functional.softplus(tensor)

    def inverse_transform(self, tensor):
        return inv_softplus(tensor)


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
        if constraint.initial_value is not None:
            self.initialize(**{param_name: constraint.inverse_transform(torch.as_tensor(constraint.initial_value))})

    def constraint_for(self, param_name: str):
        return self._constraints_map.get(param_name)

    def register_prior(self, name, prior, param_or_closure, setting_closure=None):
        if isinstance(param_or_closure, str):
            pname = param_or_closure

            def closure(m, pname=pname):
                return getattr(m, pname)
        else:
            closure = param_or_closure
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
