"""CPU: pin the oracle's covariance arithmetic against (a) outputs of the reference's own code
(tests/golden/kernel_values.npz, produced by tests/golden/make_golden.py executing
gpytorch/functions/{rbf,matern}_covariance.py and kernels/kernel.py::sq_dist/dist) and (b) the
hand-computed known answers in the reference's unit tests."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import exact_gp as OG
from oracle import kernels as OK

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "kernel_values.npz"))
CASES = "abcde"


def _t(name):
    return torch.from_numpy(Z[name])


@pytest.mark.parametrize("case", CASES)
def test_sq_dist_and_dist_bitwise(case):
    x1, x2, same = _t(f"{case}_x1"), _t(f"{case}_x2"), bool(Z[f"{case}_same"])
    assert torch.equal(OK.sq_dist(x1, x2, same), _t(f"{case}_sq_dist"))
    assert torch.equal(OK.dist(x1, x2, same), _t(f"{case}_dist"))


@pytest.mark.parametrize("case", CASES)
def test_rbf_and_matern_values_bitwise(case):
    x1, x2, ls = _t(f"{case}_x1"), _t(f"{case}_x2"), float(Z[f"{case}_ls"])
    lsz = torch.tensor([[ls]], dtype=x1.dtype)
    assert torch.equal(OK.rbf(x1, x2, lsz), _t(f"{case}_rbf"))
    for nu, key in [(0.5, "matern05"), (1.5, "matern15"), (2.5, "matern25")]:
        got = OK.matern(x1, x2, lsz, nu)
        ref = _t(f"{case}_{key}")
        # identical formula; the reference fuses in-place ops in a different order for nu=2.5
        assert torch.allclose(got, ref, rtol=1e-6 if x1.dtype == torch.float32 else 1e-13, atol=0)


@pytest.mark.parametrize("case", CASES)
def test_lengthscale_gradients(case):
    """dK/dl restatement vs the reference autograd Functions' backward (rbf_covariance.py:26-29,
    matern_covariance.py:53-56) contracted with a fixed W."""
    x1, x2, ls, W = _t(f"{case}_x1"), _t(f"{case}_x2"), float(Z[f"{case}_ls"]), _t(f"{case}_W")
    lsz = torch.tensor([[ls]], dtype=x1.dtype)
    rt = 2e-4 if x1.dtype == torch.float32 else 1e-10
    g = (OK.rbf_dl(x1, x2, lsz) * W).sum()
    assert torch.allclose(g, _t(f"{case}_rbf_dls").sum(), rtol=rt)
    for nu, key in [(0.5, "matern05"), (1.5, "matern15"), (2.5, "matern25")]:
        g = (OK.matern_dl(x1, x2, lsz, nu) * W).sum()
        assert torch.allclose(g, _t(f"{case}_{key}_dls").sum(), rtol=rt)


@pytest.mark.parametrize("case", "abc")
def test_direct_form_equals_gram_form(case):
    """The pairwise-difference form the HIP kernels use (keops/*_kernel.py) equals the Gram-trick
    form (kernels/kernel.py:26-49) to float64 rounding."""
    x1, x2, ls = _t(f"{case}_x1"), _t(f"{case}_x2"), float(Z[f"{case}_ls"])
    assert torch.allclose(OK.rbf(x1, x2, ls, direct=True), _t(f"{case}_rbf"), atol=1e-12)
    assert torch.allclose(OK.matern(x1, x2, ls, 2.5, direct=True), _t(f"{case}_matern25"), atol=1e-7)


def test_reference_known_answers():
    # test/kernels/test_rbf_kernel.py:126-142
    a = torch.tensor([4.0, 2, 8]).view(3, 1)
    b = torch.tensor([0.0, 2, 4]).view(3, 1)
    actual = torch.tensor([[16.0, 4, 0], [4, 0, 4], [64, 36, 16]]).mul_(-0.5).div_(4).exp_()
    assert torch.norm(OK.rbf(a, b, 2.0) - actual) < 1e-5
    # test/kernels/test_rbf_kernel.py:21-38 (ARD)
    a2 = torch.tensor([[1.0, 2], [2, 4]])
    b2 = torch.tensor([[1.0, 3], [0, 4]])
    lsv = torch.tensor([[1.0, 2]])
    act = ((a2 / lsv).unsqueeze(-2) - (b2 / lsv).unsqueeze(-3)).pow(2).sum(-1).mul_(-0.5).exp()
    assert torch.norm(OK.rbf(a2, b2, lsv) - act) < 1e-5
    # test/kernels/test_matern_kernel.py:41-77
    bb = torch.tensor([0.0, 2]).view(2, 1)
    dd = torch.tensor([[4.0, 2], [2, 0], [8, 6]])
    assert torch.norm(OK.matern(a, bb, 2.0, 0.5) - dd.div(-2).exp()) < 1e-3
    d3 = dd * math.sqrt(3) / 2
    assert torch.norm(OK.matern(a, bb, 2.0, 1.5) - (d3 + 1) * torch.exp(-d3)) < 1e-3
    d5 = dd * math.sqrt(5) / 2
    assert torch.norm(OK.matern(a, bb, 2.0, 2.5) - (d5**2 / 3 + d5 + 1) * torch.exp(-d5)) < 1e-3


def test_mvn_log_prob_known_answer():
    # test/distributions/test_multivariate_normal.py:28,40-43: mean [0,1,2], covariance diag [1, 0.75, 1.5]
    mean = torch.tensor([0.0, 1, 2], dtype=torch.float64)
    cov = torch.diag(torch.tensor([1.0, 0.75, 1.5], dtype=torch.float64))
    lp = OG.dense_log_prob(cov, torch.zeros(3, dtype=torch.float64) - mean)
    assert abs(float(lp) - (-4.8157)) < 1e-4


def test_parameter_transforms_golden():
    """Raw <-> constrained parameter maps against outputs of the reference's own ``gpytorch/utils/transforms.py``
    (tests/golden/transform_values.npz): inv_softplus bit for bit, the sigmoid inverse of the Interval constraint to
    rounding; softplus(inv_softplus(x)) == x; the noise lower bound of ``noise_models.py:29-30`` (GreaterThan(1e-4))."""
    import os

    import numpy as np

    from gpytorch_amd.module import GreaterThan, Interval, Positive, inv_softplus

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "transform_values.npz"))
    x = torch.from_numpy(z["x"])
    assert torch.equal(inv_softplus(x), torch.from_numpy(z["inv_softplus"]))
    assert torch.equal(Positive().inverse_transform(x), torch.from_numpy(z["inv_softplus"]))
    assert torch.allclose(Positive().transform(torch.from_numpy(z["inv_softplus"])), x, rtol=1e-9, atol=0)
    p = torch.from_numpy(z["p"])
    iv = Interval(2.0, 5.0)
    assert torch.allclose(iv.inverse_transform(2.0 + 3.0 * p), torch.from_numpy(z["inv_sigmoid"]), rtol=1e-12, atol=1e-13)
    gt = GreaterThan(1e-4)
    assert torch.allclose(gt.transform(gt.inverse_transform(x + 1e-4)), x + 1e-4, rtol=1e-9, atol=0)
    assert float(gt.transform(torch.tensor(-50.0, dtype=torch.float64))) >= 1e-4 * (1 - 1e-6)  # the bound is a float32 buffer


def test_periodic_and_rq_golden():
    """oracle.kernels.periodic / rq against the outputs of the reference's OWN PeriodicKernel.forward (periodic_kernel.py:125-142) and
    RQKernel.forward (rq_kernel.py:61-74) run through its own Kernel.covar_dist (tests/golden/composite_values.npz, make_golden.py):
    values, diagonals and the gradients of sum(W * K) w.r.t. lengthscale, period_length and alpha."""
    from oracle import kernels as OK

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "composite_values.npz"))
    for name in "pqrs":
        x1, x2 = torch.from_numpy(z[f"{name}_x1"]), torch.from_numpy(z[f"{name}_x2"])
        W = torch.from_numpy(z[f"{name}_W"])
        f32 = x1.dtype == torch.float32
        tol = 2e-5 if f32 else 1e-11
        ls = torch.from_numpy(z[f"{name}_ls"]).requires_grad_(True)
        per = torch.from_numpy(z[f"{name}_period"]).requires_grad_(True)
        al = torch.from_numpy(z[f"{name}_alpha"]).requires_grad_(True)
        same = bool(z[f"{name}_same"])
        kp = OK.periodic(x1, x2, ls, per)
        assert np.abs(kp.detach().numpy() - z[f"{name}_periodic"]).max() < tol, name
        gl, gp = torch.autograd.grad((kp * W).sum(), [ls, per])
        gtol = 2e-3 if f32 else 1e-9
        assert np.allclose(gl.numpy(), z[f"{name}_periodic_dls"], rtol=gtol, atol=gtol * np.abs(z[f"{name}_periodic_dls"]).max()), name
        assert np.allclose(gp.numpy(), z[f"{name}_periodic_dperiod"], rtol=gtol, atol=gtol * np.abs(z[f"{name}_periodic_dperiod"]).max()), name
        assert np.array_equal(z[f"{name}_periodic_diag"], np.ones_like(z[f"{name}_periodic_diag"]))
        # the reference's Gram-trick sq_dist (the oracle's default form) and the direct pairwise form both match its output
        for direct in (False, True):
            kr = OK.rq(x1, x2, ls, al, x1_eq_x2=same, direct=direct)
            assert np.abs(kr.detach().numpy() - z[f"{name}_rq"]).max() < (tol if not direct else max(tol, 1e-10) * 10), (name, direct)
        kr = OK.rq(x1, x2, ls, al, x1_eq_x2=same)
        gl, ga = torch.autograd.grad((kr * W).sum(), [ls, al])
        assert np.allclose(gl.numpy(), z[f"{name}_rq_dls"], rtol=gtol, atol=gtol * np.abs(z[f"{name}_rq_dls"]).max()), name
        assert np.allclose(ga.numpy(), z[f"{name}_rq_dalpha"], rtol=gtol, atol=gtol * abs(float(z[f"{name}_rq_dalpha"][0]))), name
        assert np.array_equal(z[f"{name}_rq_diag"], np.ones_like(z[f"{name}_rq_diag"]))
