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


@pytest.mark.parametrize("family", ["rbf", "matern32", "matern12", "matern52", "periodic", "rq", "scale_rbf"])
def test_generic_kernel_battery_on_the_cpu_double(family, monkeypatch):
    """gpytorch/test/base_kernel_test_case.py:30-197 over the kernel classes (the CPU double evaluates RQ values too; its derivatives are device-only)."""
    from tests.shim import cpu_backend

    cpu_backend.install(monkeypatch)
    import gpytorch_amd as g
    from tests.kernel_battery import families, run_battery

    (_, make, make_ard), = [f for f in families(g) if f[0] == family]
    done = run_battery(make, make_ard, torch.device("cpu"))
    assert "getitem" in done and "pickle_dtype" in done and ("ard" in done or make_ard is None)


def test_scale_kernel_unit_tests_on_the_cpu_double(monkeypatch):
    from tests.shim import cpu_backend

    cpu_backend.install(monkeypatch)
    import gpytorch_amd as g
    from tests.known_answers import check_scale_kernel_unit_tests

    check_scale_kernel_unit_tests(g, torch.device("cpu"))


def test_oracle_reproduces_the_same_literals():
    """test/kernels/test_additive_and_product_kernels.py:92-157: the four-digit literals against the oracle's RBF."""
    a = torch.tensor([4.0, 2.0, 8.0], dtype=torch.float64).view(3, 1)
    b = torch.tensor([0.0, 2.0, 2.0], dtype=torch.float64).view(3, 1)
    kd = OK.rbf(a, b, 2.0, x1_eq_x2=False).diagonal()
    for got, want in ((2 * kd, [0.2702, 2.000, 0.0222]), (3 * kd, [0.4060, 3.000, 0.0333]), (kd ** 3, [2.4788e-03, 1.000, 1.3710e-06]), (kd ** 2, [1.8316e-02, 1.000, 1.2341e-04])):
        assert float((got - torch.tensor(want, dtype=torch.float64)).norm()) < 1e-3, (got, want)


def test_rq_kernel_closed_forms_on_the_cpu_double(monkeypatch):
    """test/kernels/test_rq_kernel.py:19-126, 206-223: the rational-quadratic closed form (1 + d^2 / (2 alpha l^2))^-alpha, ARD, ARD in a batch,
    ``last_dim_is_batch`` with per-dimension lengthscales, ``initialize`` of lengthscale / alpha."""
    from tests.shim import cpu_backend

    cpu_backend.install(monkeypatch)
    import gpytorch_amd as g

    RQ = g.kernels.RQKernel
    dn = lambda x: float(x.norm())  # noqa: E731
    with torch.no_grad():
        a, b = torch.tensor([4.0, 2, 8]).view(3, 1), torch.tensor([0.0, 2]).view(2, 1)
        k = RQ().initialize(lengthscale=2.0)
        k.initialize(alpha=3.0)
        k.eval()
        actual = (1 + torch.tensor([[16.0, 4], [4, 0], [64, 36]]).div(4.0) / 6.0).pow(-3.0)
        assert dn(k(a, b).to_dense() - actual) < 1e-5
        cases = [
            (torch.tensor([[1.0, 2], [2, 4]]), torch.tensor([[1.0, 3], [0, 4]]), torch.tensor([1.0, 2]).view(1, 2), {"ard_num_dims": 2}),
            (torch.tensor([[[1.0, 2, 3], [2, 4, 0]], [[-1, 1, 2], [2, 1, 4]]]), torch.tensor([[[1.0, 3, 1]], [[2, -1, 0]]]).repeat(1, 2, 1),
             torch.tensor([[[1.0, 2, 1]]]), {"ard_num_dims": 3, "batch_shape": torch.Size([2])}),
        ]
        for a, b, ls, kw in cases:
            k = RQ(**kw)
            k.initialize(lengthscale=ls)
            k.initialize(alpha=3.0)
            k.eval()
            sa, sb = a / ls, b / ls
            actual = (1 + (sa.unsqueeze(-2) - sb.unsqueeze(-3)).pow(2).sum(-1) / 6).pow(-3)
            assert dn(k(a, b).to_dense() - actual) < 1e-5
            assert dn(k(a, b).diagonal(dim1=-1, dim2=-2) - actual.diagonal(dim1=-1, dim2=-2)) < 1e-5
            per_dim = (1 + (sa.mT.unsqueeze(-1) - sb.mT.unsqueeze(-2)).pow(2) / 6).pow(-3)
            assert dn(k(a, b, last_dim_is_batch=True).to_dense() - per_dim) < 1e-5
    k = RQ()
    k.initialize(lengthscale=3.14)
    assert abs(k.lengthscale.item() - 3.14) < 1e-5
    k = RQ(batch_shape=torch.Size([2]))
    ls = torch.tensor([3.14, 4.13]).view(2, 1, 1)
    k.initialize(lengthscale=ls)
    assert torch.allclose(k.lengthscale, ls)
    k = RQ()
    k.initialize(alpha=3.0)
    assert abs(float(k.alpha) - 3.0) < 1e-5
