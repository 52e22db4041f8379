"""``linear_operator.operators`` of the test shim: the classes of ``gpytorch_amd.operators`` under the names the reference imports."""
from gpytorch_amd import operators as _o

LinearOperator = _o.LinearOperator
DenseLinearOperator = _o.DenseLinearOperator
DiagLinearOperator = _o.DiagLinearOperator
ConstantDiagLinearOperator = _o.ConstantDiagLinearOperator
RootLinearOperator = _o.RootLinearOperator
MatmulLinearOperator = _o.MatmulLinearOperator
SumLinearOperator = _o.SumLinearOperator
ZeroLinearOperator = _o.ZeroLinearOperator
to_dense = _o.to_dense
to_linear_operator = _o.to_linear_operator

_PLACEHOLDERS = {}


def _placeholder(name):
    """A class that exists (the reference subclasses / isinstance-checks it at import time) and refuses to be instantiated."""
    if name not in _PLACEHOLDERS:
        def __init__(self, *args, **kwargs):
            raise NotImplementedError(f"linear_operator shim: {name} is off the exact-GP path and not provided")

        _PLACEHOLDERS[name] = type(name, (LinearOperator,), {"__init__": __init__, "__module__": __name__})
    return _PLACEHOLDERS[name]


def __getattr__(name):   # PEP 562: `from linear_operator.operators import XLinearOperator` for anything not mapped above
    if name.endswith("LinearOperator"):
        return _placeholder(name)
    raise AttributeError(name)
