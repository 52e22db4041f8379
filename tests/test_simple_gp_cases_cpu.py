"""test/examples/test_simple_gp_regression.py:47-330 over the product's layers on the CPU double (host logic: prior mode, recursive ``initialize``,
interpolating posterior at tiny lengthscale / noise, ``skip_posterior_variances``, a single training point, 50 Adam steps -> MAE < 0.05, fantasy
updates with gradients to the fantasy inputs).  Device twin: tests/test_gpu_reference_examples.py."""
import pytest
import torch

from tests import simple_gp_cases as C


@pytest.mark.parametrize("case", C.CASES, ids=[c.__name__ for c in C.CASES])
@pytest.mark.filterwarnings("ignore")
def test_simple_gp_regression_case_on_the_cpu_double(case, monkeypatch):
    from tests.shim import cpu_backend

    cpu_backend.install(monkeypatch)
    import gpytorch_amd as g

    torch.manual_seed(1)
    case(g, torch.device("cpu"))
