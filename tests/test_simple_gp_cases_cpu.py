"""test/examples/test_simple_gp_regression.py:47-330 over the product's layers on the CPU double (host logic: prior mode, recursive ``initialize``,
interpolating posterior at tiny lengthscale / noise, ``skip_posterior_variances``, a single training point, 50 Adam steps -> MAE < 0.05, fantasy
updates with gradients to the fantasy inputs), test_fixed_noise_fanatasy_updates.py, test_missing_data.py (mask / fill policies: single, batch,
multitask), test_batch_gp_regression.py, test_kronecker_multitask_gp_regression.py.  Device twin: tests/test_gpu_reference_examples.py."""
import pytest
import torch

from tests import simple_gp_cases as C


ALL = C.CASES + C.MULTITASK_CASES


@pytest.mark.parametrize("case", ALL, ids=[c.__name__ for c in ALL])
@pytest.mark.filterwarnings("ignore")
def test_simple_gp_regression_case_on_the_cpu_double(case, monkeypatch):
    from tests.shim import cpu_backend

    cpu_backend.install(monkeypatch)
    import gpytorch_amd as g

    torch.manual_seed(1)
    case(g, torch.device("cpu"))
