"""Host logic of the fixed-noise likelihoods, following the reference's own unit tests step by step (``test/likelihoods/test_gaussian_likelihood.py:56-95``
``test_fixed_noise_gaussian_likelihood``, ``:129-160`` ``test_dirichlet_classification_likelihood``; ``gaussian_likelihood.py:283-296`` for the batch
shape of the learned noise).  No kernel evaluation: diagonal operators only, so no native library is needed."""
import os
import warnings

import numpy as np
import pytest
import torch

import gpytorch_amd as g
from gpytorch_amd.likelihoods import DirichletClassificationLikelihood, FixedGaussianNoise, FixedNoiseGaussianLikelihood
from gpytorch_amd.operators import DiagLinearOperator

MVN = g.distributions.MultivariateNormal


@pytest.mark.parametrize("dtype", [torch.float, torch.double])
def test_fixed_noise_gaussian_likelihood(dtype):
    torch.manual_seed(0)
    noise = 0.1 + torch.rand(4, dtype=dtype)
    lkhd = FixedNoiseGaussianLikelihood(noise=noise)
    assert isinstance(lkhd.noise_covar, FixedGaussianNoise)
    assert torch.equal(noise, lkhd.noise)
    new_noise = 0.1 + torch.rand(4, dtype=dtype)
    lkhd.noise = new_noise
    assert torch.equal(lkhd.noise, new_noise)
    out = lkhd(MVN(torch.zeros(4, dtype=dtype), DiagLinearOperator(torch.ones(4, dtype=dtype))))
    assert torch.allclose(out.variance, 1 + new_noise)
    mvn5 = MVN(torch.zeros(5, dtype=dtype), DiagLinearOperator(torch.ones(5, dtype=dtype)))
    with pytest.warns(UserWarning):          # sizes differ and no noise given: a warned no-op
        lkhd(mvn5)
    obs_noise = 0.1 + torch.rand(5, dtype=dtype)
    out = lkhd(mvn5, noise=obs_noise)
    assert torch.allclose(out.variance, 1 + obs_noise)
    floor = g.settings.min_fixed_noise.value(dtype)
    noise[:2] = 0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lkhd = FixedNoiseGaussianLikelihood(noise=noise)
    expected = noise.clone()
    expected[:2] = floor
    assert torch.allclose(lkhd.noise, expected)


@pytest.mark.parametrize("dtype", [torch.float, torch.double])
def test_dirichlet_classification_likelihood(dtype):
    torch.manual_seed(1)
    labels = (torch.rand(6, dtype=dtype) > 0.5).long()
    labels[0], labels[1] = 0, 1                                   # (both classes present: the class count is read from the labels)
    lkhd = DirichletClassificationLikelihood(labels, dtype=dtype)
    assert isinstance(lkhd.noise_covar, FixedGaussianNoise)
    assert lkhd.num_classes == 2 and lkhd.transformed_targets.shape == (2, 6) and lkhd.noise_covar.noise.shape == (2, 6)
    labels2 = (torch.rand(6, dtype=dtype) > 0.5).long()
    labels2[0], labels2[1] = 1, 0
    new_noise, _, _ = lkhd._prepare_targets(labels2, dtype=dtype)
    lkhd.noise = new_noise
    assert torch.equal(lkhd.noise, new_noise)
    out = lkhd(MVN(torch.zeros(6, dtype=dtype), DiagLinearOperator(torch.ones(6, dtype=dtype))))
    assert torch.allclose(out.variance, 1 + new_noise)
    mvn5 = MVN(torch.zeros(5, dtype=dtype), DiagLinearOperator(torch.ones(5, dtype=dtype)))
    with pytest.warns(UserWarning):
        lkhd(mvn5)
    obs = torch.tensor([0, 1, 1, 0, 1])
    out = lkhd(mvn5, targets=obs)
    assert torch.allclose(out.variance, 1.0 + lkhd._prepare_targets(obs, dtype=dtype)[0])


def test_dirichlet_transform_equals_the_references_own_output():
    """tests/golden/reference_notebook_runs.npz holds what the reference's ``_prepare_targets`` returned for the notebook's 500 labels (executed by
    tests/golden/make_notebook_golden.py)."""
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_notebook_runs.npz"))
    lkhd = DirichletClassificationLikelihood(torch.from_numpy(G["cls_labels"]), learn_additional_noise=True)
    assert lkhd.num_classes == 3
    assert np.abs(lkhd.noise_covar.noise.numpy() - G["cls_fixed_noise"]).max() < 1e-6
    assert np.abs(lkhd.transformed_targets.numpy() - G["cls_targets"]).max() < 1e-6
    # one learned noise per member (gaussian_likelihood.py:291-296), at softplus(0) + 1e-4
    assert lkhd.second_noise_covar.noise.shape == (3, 1)
    assert torch.allclose(lkhd.second_noise_covar.noise, torch.full((3, 1), 0.6931472 + 1e-4))
    plain = FixedNoiseGaussianLikelihood(noise=torch.rand(3, 5) + 0.1, learn_additional_noise=True, batch_shape=torch.Size([3]))
    assert plain.second_noise_covar.raw_noise.shape == (3, 1)
    assert FixedNoiseGaussianLikelihood(noise=torch.rand(5) + 0.1, learn_additional_noise=True).second_noise_covar.raw_noise.shape == (1,)


@pytest.mark.parametrize("interleaved,lazy", [(True, True), (True, False), (False, True)])
def test_multitask_likelihood_marginal_variance(interleaved, lazy):
    """test/likelihoods/test_multitask_gaussian_likelihood.py:31-37 (rank 0; both flattenings of the multitask covariance, :76-82)."""
    from gpytorch_amd.operators import DenseLinearOperator

    lik = g.likelihoods.MultitaskGaussianLikelihood(num_tasks=4, rank=0, has_global_noise=False)
    lik.task_noises = torch.tensor([0.1, 0.2, 0.3, 0.4])
    c = torch.tensor([1, 0.6, 0.4, 0.2, 0.1])
    i = torch.arange(5)
    data = c[(i[:, None] - i[None, :]).abs()]                       # the Toeplitz data covariance of the reference's test
    t = torch.tensor([[1.0], [2.0], [3.0], [4.0]])
    cov = torch.kron(data, t @ t.t()) if interleaved else torch.kron(t @ t.t(), data)
    dist = g.distributions.MultitaskMultivariateNormal(torch.randn(5, 4), DenseLinearOperator(cov) if lazy else cov, interleaved=interleaved)
    torch.testing.assert_close(lik(dist).variance, torch.tensor([1.1, 4.2, 9.3, 16.4]).repeat(5, 1))


def test_multitask_likelihood_setters():
    """test/likelihoods/test_multitask_gaussian_likelihood.py:47-62 (rank 0: diagonal task noises; a full task-noise matrix cannot be set)."""
    lik = g.likelihoods.MultitaskGaussianLikelihood(num_tasks=3, rank=0)
    lik.noise = 0.5
    assert abs(lik.noise.item() - 0.5) < 1e-6
    lik.task_noises = torch.tensor([0.04, 0.04, 0.04])
    assert all(abs(lik.task_noises[k].item() - 0.04) < 1e-6 for k in range(3))
    a = torch.randn(3, 2)
    with pytest.raises(AttributeError, match="task noises"):
        lik.task_noise_covar = a @ a.t()
    with pytest.raises(NotImplementedError):                          # (rank > 0 -- inter-task noise correlations -- is outside the built path, said loudly)
        g.likelihoods.MultitaskGaussianLikelihood(num_tasks=3, rank=2)
