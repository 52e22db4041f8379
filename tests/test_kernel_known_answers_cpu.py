"""Composed-kernel known answers of the reference's unit tests through the product's kernel classes on the CPU double (host logic: sums, products of
squared-exponential members as one feature-map operator, batch broadcasting, the elementwise ``diag=True``); device twin: tests/test_gpu_compose.py."""
import pytest
import torch

from oracle import kernels as OK


def test_sum_and_product_known_answers_on_the_cpu_double(monkeypatch):
    from tests.shim import cpu_backend

    cpu_backend.install(monkeypatch)
    import gpytorch_amd as g
    from tests.known_answers import check_sum_product_known_answers

    check_sum_product_known_answers(g, torch.device("cpu"))


def test_stationary_and_periodic_unit_tests_on_the_cpu_double(monkeypatch):
    from tests.shim import cpu_backend

    cpu_backend.install(monkeypatch)
    import gpytorch_amd as g
    from tests.known_answers import check_stationary_and_periodic_unit_tests

    check_stationary_and_periodic_unit_tests(g, torch.device("cpu"))


@pytest.mark.parametrize("family", ["rbf", "matern32", "matern12", "matern52", "periodic", "scale_rbf"])
def test_generic_kernel_battery_on_the_cpu_double(family, monkeypatch):
    """gpytorch/test/base_kernel_test_case.py:30-197 over the kernel classes (the CPU double has no RQ: that family runs on the device only)."""
    from tests.shim import cpu_backend

    cpu_backend.install(monkeypatch)
    import gpytorch_amd as g
    from tests.kernel_battery import families, run_battery

    (_, make, make_ard), = [f for f in families(g) if f[0] == family]
    done = run_battery(make, make_ard, torch.device("cpu"))
    assert "getitem" in done and "pickle_dtype" in done and ("ard" in done or make_ard is None)


def test_oracle_reproduces_the_same_literals():
    """test/kernels/test_additive_and_product_kernels.py:92-157: the four-digit literals against the oracle's RBF."""
    a = torch.tensor([4.0, 2.0, 8.0], dtype=torch.float64).view(3, 1)
    b = torch.tensor([0.0, 2.0, 2.0], dtype=torch.float64).view(3, 1)
    kd = OK.rbf(a, b, 2.0, x1_eq_x2=False).diagonal()
    for got, want in ((2 * kd, [0.2702, 2.000, 0.0222]), (3 * kd, [0.4060, 3.000, 0.0333]), (kd ** 3, [2.4788e-03, 1.000, 1.3710e-06]), (kd ** 2, [1.8316e-02, 1.000, 1.2341e-04])):
        assert float((got - torch.tensor(want, dtype=torch.float64)).norm()) < 1e-3, (got, want)
