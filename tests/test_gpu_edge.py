"""GPU: edge cases and error behaviour of the hot path, modelled on what the reference's unit tests poke at
(test/kernels/test_rbf_kernel.py: ARD / active_dims / 1-D inputs; linear_cg's NaN check and non-convergence
warning; settings.skip_posterior_variances / skip_logdet_forward; duplicate inputs)."""
import math
import warnings

import pytest
import torch

from oracle import exact_gp as OG
from oracle import kernels as OK
from tests.util import make_data, rel_err

pytestmark = pytest.mark.gpu


def test_duplicate_points_and_extreme_lengthscales(dev):
    """Exact duplicates (singular K, regularised only by the noise) and the K ~ 11^T / K ~ I regimes."""
    from gpytorch_amd import backend as B
    from gpytorch_amd.linear_cg import linear_cg

    n = 1200
    X, y = make_data(n // 2, 3)
    X = torch.cat([X, X], 0)  # every point twice
    rhs = torch.randn(n, 4, generator=torch.Generator().manual_seed(0), dtype=torch.float64)
    sc, s2 = torch.tensor([1.0], device=dev), torch.tensor([0.1], device=dev)
    for ls in (0.3, 25.0, 0.004):
        xp = B.prep_points("rbf", X.float().to(dev), torch.tensor(ls), X.mean(0).to(dev))
        K = OK.rbf(X, X, ls, direct=True)
        out = B.from_probe_major(B.kv(xp, xp, B.to_probe_major(rhs.to(dev))), n)
        assert rel_err(out, K @ rhs) < 5e-5, ls
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sol_t, info = linear_cg(xp, sc, s2, B.to_probe_major(rhs.to(dev)), tolerance=1e-4, max_iter=400)
        ref = torch.linalg.solve(K + 0.1 * torch.eye(n, dtype=torch.float64), rhs)
        assert rel_err(B.from_probe_major(sol_t, n), ref) < 2e-3, ls


def test_nan_rhs_raises_and_nonconvergence_warns(dev):
    from gpytorch_amd import backend as B
    from gpytorch_amd.linear_cg import NumericalWarning, linear_cg

    n = 900
    X, y = make_data(n, 3)
    xp = B.prep_points("rbf", X.float().to(dev), torch.tensor(0.25))
    sc, s2 = torch.tensor([1.0], device=dev), torch.tensor([0.1], device=dev)
    bad = y.clone()
    bad[17] = float("nan")
    with pytest.raises(RuntimeError, match="NaN"):
        linear_cg(xp, sc, s2, B.to_probe_major(bad.unsqueeze(-1).to(dev)), tolerance=1e-3, max_iter=50)
    with pytest.warns(NumericalWarning, match="CG terminated"):
        _, info = linear_cg(xp, sc, s2, B.to_probe_major(y.unsqueeze(-1).to(dev)), tolerance=1e-9, max_iter=12)
    assert info.iterations == 12 and not info.tolerance_reached


def test_kernel_call_semantics_ard_active_dims_1d(dev):
    """Kernel.__call__ (kernels/kernel.py:454-534): 1-D inputs are promoted to [n, 1], active_dims select
    columns, ARD lengthscales scale per dimension, diag=True returns the diagonal tensor."""
    import gpytorch_amd as g

    a = torch.tensor([4.0, 2, 8], device=dev)
    b = torch.tensor([0.0, 2, 4], device=dev)
    k = g.kernels.RBFKernel().to(dev)
    k.lengthscale = 2.0
    actual = torch.tensor([[16.0, 4, 0], [4, 0, 4], [64, 36, 16]]).mul_(-0.5).div_(4).exp_()  # test_rbf_kernel.py:126-142
    assert torch.norm(k(a, b).to_dense().cpu() - actual) < 1e-5
    assert torch.norm(k(a, b, diag=True).cpu() - actual.diagonal()) < 1e-5
    a2 = torch.stack([a, torch.tensor([1.0, 2, 3], device=dev)], 1)
    k2 = g.kernels.RBFKernel(active_dims=[0]).to(dev)
    k2.lengthscale = 2.0
    assert torch.norm(k2(a2, b.unsqueeze(-1).expand(3, 2).contiguous()).to_dense().cpu() - actual) < 1e-5  # :105-124
    aa = torch.tensor([[1.0, 2], [2, 4]], device=dev)
    bb = torch.tensor([[1.0, 3], [0, 4]], device=dev)
    k3 = g.kernels.RBFKernel(ard_num_dims=2).to(dev)
    k3.lengthscale = torch.tensor([[1.0, 2.0]])
    ls = torch.tensor([[1.0, 2.0]])
    act = ((aa.cpu() / ls).unsqueeze(-2) - (bb.cpu() / ls).unsqueeze(-3)).pow(2).sum(-1).mul_(-0.5).exp()  # :21-38
    assert torch.norm(k3(aa, bb).to_dense().cpu() - act) < 1e-5
    with pytest.raises(RuntimeError):
        k3(torch.rand(4, 3, device=dev))


def test_ard_mll_gradients(dev):
    """Per-dimension lengthscale gradients through the model API (Cholesky branch -> fused ARD derivative)."""
    import gpytorch_amd as g

    n, d = 300, 4
    X, y = make_data(n, d)
    ls = torch.tensor([[0.3, 0.5, 0.8, 1.2]], dtype=torch.float64)

    class M(g.models.ExactGP):
        def __init__(self, x, yy, lik):
            super().__init__(x, yy, lik)
            self.mean_module = g.means.ZeroMean()
            self.covar_module = g.kernels.ScaleKernel(g.kernels.MaternKernel(nu=2.5, ard_num_dims=d))

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    lik = g.likelihoods.GaussianLikelihood().to(dev)
    m = M(X.float().to(dev), y.float().to(dev), lik).to(dev)
    m.covar_module.base_kernel.lengthscale = ls.float()
    m.covar_module.outputscale = 1.4
    lik.noise = 0.2
    mll = g.ExactMarginalLogLikelihood(lik, m)
    m.train(); lik.train()
    val = mll(m(m.train_inputs[0]), m.train_targets)
    val.backward()
    lsr = ls.clone().requires_grad_(True)
    ref = OG.dense_mll("matern52", X, y, lsr, 1.4, 0.2)
    (gl,) = torch.autograd.grad(ref, lsr)
    assert abs(float(val) - float(ref)) < 2e-4
    got = m.covar_module.base_kernel.raw_lengthscale.grad.double().cpu()
    exp = gl * (1 - torch.exp(-ls))
    assert rel_err(got, exp) < 3e-3, (got, exp)


def test_settings_skip_flags(dev):
    import gpytorch_amd as g

    n, ns = 1000, 20
    X, y = make_data(n, 3)
    Xs, _ = make_data(ns, 3, seed=2)
    lik = g.likelihoods.GaussianLikelihood().to(dev)

    class M(g.models.ExactGP):
        def __init__(self, x, yy, l):
            super().__init__(x, yy, l)
            self.mean_module = g.means.ConstantMean()
            self.covar_module = g.kernels.ScaleKernel(g.kernels.RBFKernel())

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    m = M(X.float().to(dev), y.float().to(dev), lik).to(dev)
    m.covar_module.base_kernel.lengthscale = 0.25
    S = g.settings
    m.eval(); lik.eval()
    with torch.no_grad(), S.max_cholesky_size(0), S.skip_posterior_variances():
        pred = m(Xs.float().to(dev))
        assert float(pred.lazy_covariance_matrix.to_dense().abs().max()) == 0.0
        assert pred.mean.shape == (ns,)
    m.train(); lik.train()
    mll = g.ExactMarginalLogLikelihood(lik, m)
    with S.max_cholesky_size(0), S.skip_logdet_forward():
        v = mll(m(m.train_inputs[0]), m.train_targets)
    with S.max_cholesky_size(0):
        v_full = mll(m(m.train_inputs[0]), m.train_targets)
    assert math.isfinite(float(v)) and float(v) != float(v_full)  # value without the log-det term
    # train-mode input check (exact_gp.py:276-280)
    with pytest.raises(RuntimeError, match="train on the training inputs"):
        m(Xs.float().to(dev))


def test_rccl_communicator_entry_points_on_a_single_rank_communicator(dev):
    """The RCCL-communicator variants of the C ABI (include/gpamd.h: gpamd_allreduce_sum_f32, gpamd_cg_stop_comm_f32) with a REAL ncclComm_t:
    a one-rank communicator created through librccl (ncclGetUniqueId / ncclCommInitRank) -- the all-reduce is then the identity, which checks
    the handle passing, the datatype / op constants and the stream ordering; multi-rank behaviour is RCCL's."""
    import ctypes as C

    from gpytorch_amd import backend as B
    from gpytorch_amd._lib import check, lib

    try:
        rccl = C.CDLL("librccl.so")
    except OSError:
        pytest.skip("librccl.so not present")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        buf = torch.arange(1, 9, device=dev, dtype=torch.float32)
        want = buf.clone()
        check(lib().gpamd_allreduce_sum_f32(B._ptr(buf), buf.numel(), comm, B._stream(dev)), "allreduce")
        torch.cuda.synchronize(dev)
        assert torch.equal(buf, want)
        assert lib().gpamd_allreduce_sum_f32(None, 4, comm, B._stream(dev)) == -1
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
