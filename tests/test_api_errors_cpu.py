"""Error behaviour of the host-side mirror, without a GPU: the same exceptions, at the same points and with the same
messages, as the reference raises before any arithmetic happens (exact_gp.py:113-149,265-280; kernel.py:492-510;
matern_kernel.py:80-82; index_kernel.py:60-62; likelihood.py; multivariate_normal.py) -- and the product's own rule
that nothing runs without the HIP library and a ROCm device (no CPU fallback)."""
import pytest
import torch

import gpytorch_amd as g


class _GP(g.models.ExactGP):
    def __init__(self, x, y, lik):
        super().__init__(x, y, lik)
        self.mean_module = g.means.ConstantMean()
        self.covar_module = g.kernels.ScaleKernel(g.kernels.RBFKernel(ard_num_dims=2))

    def forward(self, x):
        return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))


def test_exact_gp_argument_errors():
    lik = g.likelihoods.GaussianLikelihood()
    with pytest.raises(RuntimeError, match="Train inputs must be a tensor"):
        _GP([1.0, 2.0], torch.zeros(2), lik)
    m = _GP(None, None, lik)
    m.train()
    with pytest.raises(RuntimeError, match="cannot be None in training mode"):
        m(torch.zeros(3, 2))
    m = _GP(torch.zeros(5, 2), torch.zeros(5), lik)
    with pytest.raises(RuntimeError, match="Cannot modify shape/dtype/device of train inputs"):
        m.set_train_data(torch.zeros(6, 2), torch.zeros(6), strict=True)
    with pytest.raises(RuntimeError, match="Cannot modify shape of train targets"):
        m.set_train_data(targets=torch.zeros(6), strict=True)
    m.set_train_data(torch.zeros(6, 2), torch.zeros(6), strict=False)
    assert m.train_inputs[0].shape == (6, 2) and m.prediction_strategy is None
    m.eval()
    with pytest.raises(RuntimeError, match="Fantasy observations can only be added after making predictions"):
        m.get_fantasy_model(torch.zeros(2, 2), torch.zeros(2))
    # 1-D inputs are promoted to [n, 1] (exact_gp.py:60-61)
    assert g.models.ExactGP(torch.zeros(4), torch.zeros(4), lik).train_inputs[0].shape == (4, 1)


def test_kernel_and_likelihood_argument_errors():
    with pytest.raises(RuntimeError, match="nu expected to be 0.5, 1.5, or 2.5"):
        g.kernels.MaternKernel(nu=1.0)
    k = g.kernels.RBFKernel(ard_num_dims=3)
    with pytest.raises(RuntimeError, match="Expected the input to have 3 dimensionality"):
        k(torch.zeros(4, 2))
    assert g.kernels.RBFKernel(batch_shape=torch.Size([2])).raw_lengthscale.shape == (2, 1, 1)   # kernel.py:163-208
    assert g.kernels.ScaleKernel(g.kernels.RBFKernel(), batch_shape=torch.Size([3])).raw_outputscale.shape == (3,)
    with pytest.raises(RuntimeError, match="larger than the number of tasks"):
        g.kernels.IndexKernel(num_tasks=2, rank=3)
    with pytest.raises(RuntimeError, match="expects a MultivariateNormal"):
        g.likelihoods.GaussianLikelihood()(torch.zeros(3))
    with pytest.raises(RuntimeError, match="sizes do not match"):
        g.distributions.MultivariateNormal(torch.zeros(3), torch.eye(4))
    # constraints: the noise lower bound of noise_models.py:29-30 is enforced through the transform
    lik = g.likelihoods.GaussianLikelihood()
    lik.noise = 0.5
    assert abs(float(lik.noise) - 0.5) < 1e-6
    assert float(lik.noise_covar.raw_noise_constraint.lower_bound) == pytest.approx(1e-4)


def test_no_cpu_fallback():
    """The fused operators refuse host tensors loudly instead of computing on the CPU."""
    from gpytorch_amd import backend as B

    with pytest.raises(Exception) as ei:
        B.prep_points("rbf", torch.zeros(4, 2), torch.tensor(1.0))
    assert "ROCm" in str(ei.value) or "cuda" in str(ei.value).lower() or "device" in str(ei.value).lower()
    k = g.kernels.ScaleKernel(g.kernels.RBFKernel())
    op = k(torch.rand(5, 2))  # lazily evaluated: constructing the operator is fine ...
    with pytest.raises(Exception):
        op.to_dense()            # ... evaluating it is not


def test_split_contraction_setting_scopes_and_env(monkeypatch):
    """settings.split_contraction: on by default, nests like the reference's feature flags, GPAMD_KV_SPLIT=0 turns the process
    default off (read at import), and it only ever adds KV_SPLIT on top of the Gram-form flag."""
    import subprocess
    import sys

    from gpytorch_amd import backend as B
    from gpytorch_amd import settings

    assert settings.split_contraction.on()
    with settings.split_contraction(False):
        assert settings.split_contraction.off()
        with settings.split_contraction(True):
            assert settings.split_contraction.on()
        assert settings.split_contraction.off()
    assert settings.split_contraction.on()
    code = "import gpytorch_amd.settings as s; print(int(s.split_contraction.on()))"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env={**__import__("os").environ, "GPAMD_KV_SPLIT": "0"})
    assert out.stdout.strip() == "0", out.stderr
    assert B.KV_SPLIT == 8 and B.KV_GRAM == 1


def test_prior_mode_setting():
    """``settings.prior_mode`` (gpytorch/settings.py:336-344; models/exact_gp.py:285): an ExactGP WITH training data evaluates the prior
    at the given inputs -- no prediction strategy is built, the mean is the prior mean, the covariance the lazy prior kernel operator."""
    lik = g.likelihoods.GaussianLikelihood()
    x, y = torch.rand(7, 2), torch.randn(7)
    m = _GP(x, y, lik)
    m.mean_module.constant = 0.3
    m.eval()
    xs = torch.rand(4, 2)
    with g.settings.prior_mode(True):
        out = m(xs)
    assert m.prediction_strategy is None
    assert out.mean.shape == (4,) and torch.allclose(out.mean, torch.full((4,), 0.3))
    assert tuple(out.lazy_covariance_matrix.shape) == (4, 4)
    assert g.settings.prior_mode.off()


def test_functional_aliases_on_dense_operands():
    """The functional surface of ``gpytorch/__init__.py:34-278`` on plain tensors (wrapped as dense operators, as the reference's
    ``to_linear_operator`` does): solve / inv_quad / logdet / inv_quad_logdet against torch.linalg, and the two deprecated aliases the
    reference still exports (``matmul``, ``inv_matmul``) with their DeprecationWarning."""
    gen = torch.Generator().manual_seed(0)
    A = torch.randn(9, 9, generator=gen, dtype=torch.float64)
    A = A @ A.t() + torch.eye(9, dtype=torch.float64)
    R = torch.randn(9, 3, generator=gen, dtype=torch.float64)
    Lm = torch.randn(2, 9, generator=gen, dtype=torch.float64)
    assert torch.allclose(g.solve(A, R), torch.linalg.solve(A, R))
    assert torch.allclose(g.solve(A, R, Lm), Lm @ torch.linalg.solve(A, R))
    assert torch.allclose(g.inv_quad(A, R), (R * torch.linalg.solve(A, R)).sum())
    assert torch.allclose(g.logdet(A), torch.logdet(A))
    iq, ld = g.inv_quad_logdet(A, R, logdet=True, reduce_inv_quad=False)
    assert torch.allclose(iq, (R * torch.linalg.solve(A, R)).sum(0)) and torch.allclose(ld, torch.logdet(A))
    with pytest.warns(DeprecationWarning, match="Use torch.matmul"):
        assert torch.allclose(g.matmul(A, R), A @ R)
    with pytest.warns(DeprecationWarning, match="Use gpytorch.solve"):
        assert torch.allclose(g.inv_matmul(A, R), torch.linalg.solve(A, R))


def test_multi_device_kernel_constructor_compatibility():
    """``gpytorch.kernels.MultiDeviceKernel(base_kernel, device_ids, output_device)`` (multi_device_kernel.py:14-47) constructs and
    delegates: same lazy operator as the wrapped kernel, ``base_kernel`` property, ``module.``-prefixed state dict, a warning when one
    process is handed several devices (multi-GPU here is one process per GPU + settings.sharding)."""
    base = g.kernels.ScaleKernel(g.kernels.RBFKernel(ard_num_dims=2))
    with pytest.warns(RuntimeWarning, match="one process per GPU"):
        k = g.kernels.MultiDeviceKernel(base, device_ids=[torch.device("cpu"), torch.device("cpu")], output_device=torch.device("cpu"))
    assert k.base_kernel is base and k.output_device == torch.device("cpu")
    keys = sorted(k.state_dict())
    assert all(key.startswith("module.") for key in keys) and {"module.base_kernel.raw_lengthscale", "module.raw_outputscale"} <= set(keys)
    x = torch.rand(6, 2)
    op, ref = k(x), base(x)
    assert type(op) is type(ref) and tuple(op.shape) == (6, 6)
    assert torch.equal(k(x, diag=True), base(x, diag=True))
    assert g.settings.sharding.probe_group() is None          # no process group: nothing installed
