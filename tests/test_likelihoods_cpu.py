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
