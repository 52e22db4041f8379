"""``gpytorch.constraints``: the parameter constraints, under the reference's import path (``gpytorch/constraints/__init__.py``)."""
from .module import GreaterThan, Interval, LessThan, Positive

__all__ = ["GreaterThan", "Interval", "LessThan", "Positive"]
