"""``test/distributions/test_multitask_multivariate_normal.py:42-305`` restated over ``gpytorch_amd.distributions.MultitaskMultivariateNormal``: the known
answers (entropy 11.80326, log-probability -14.52826), both flattenings, batches, and the constructors from single-output distributions."""
import math

import pytest
import torch

import gpytorch_amd as g
from gpytorch_amd.operators import DiagLinearOperator

MVN = g.distributions.MultivariateNormal
MT = g.distributions.MultitaskMultivariateNormal


@pytest.mark.parametrize("dt", [torch.float, torch.double])
def test_multitask_multivariate_normal(dt):
    mean = torch.tensor([[0, 1], [2, 3], [4, 5]], dtype=dt)
    var = torch.tensor([[1, 2], [3, 4], [5, 6]], dtype=dt)
    cov = var.view(-1).diag_embed()                                   # interleaved
    m = MT(mean, cov)
    assert torch.equal(m.mean, mean) and torch.allclose(m.variance, var) and torch.allclose(m.scale_tril, cov.sqrt())
    assert m.event_shape == torch.Size([3, 2]) and m.batch_shape == torch.Size()
    for new, mu, c in ((m + 1, m.mean + 1, cov), (m * 2, m.mean * 2, cov * 4), (m / 2, m.mean / 2, cov / 4)):
        assert isinstance(new, MT) and torch.equal(new.mean, mu) and torch.equal(new.covariance_matrix, c)
    assert abs(m.entropy().item() - 11.80326) < 1e-4
    assert abs(m.log_prob(torch.zeros(3, 2, dtype=dt)).item() + 14.52826) < 1e-4
    assert torch.allclose(m.log_prob(torch.zeros(2, 3, 2, dtype=dt)), -14.52826 * torch.ones(2, dtype=dt))
    lo, up = m.confidence_region()
    assert torch.allclose(lo, m.mean - 2 * m.stddev) and torch.allclose(up, m.mean + 2 * m.stddev)
    assert m.sample().shape == (3, 2) and m.sample(torch.Size([3])).shape == (3, 3, 2) and m.sample(torch.Size([3, 4])).shape == (3, 4, 3, 2)
    cov = var.mT.reshape(-1).diag_embed()                             # non-interleaved (task-major)
    m = MT(mean, cov, interleaved=False)
    assert torch.equal(m.mean, mean) and torch.allclose(m.variance, var) and m.event_shape == torch.Size([3, 2])
    assert abs(m.log_prob(torch.zeros(3, 2, dtype=dt)).item() + 14.52826) < 1e-4


@pytest.mark.parametrize("dt", [torch.float, torch.double])
def test_multitask_multivariate_normal_batch(dt):
    mean = torch.tensor([[0, 1], [2, 3], [4, 5]], dtype=dt).repeat(2, 1, 1)
    var = torch.tensor([[1, 2], [3, 4], [5, 6]], dtype=dt).repeat(2, 1, 1)
    m = MT(mean, var.view(2, -1).diag_embed())
    assert torch.equal(m.mean, mean) and torch.allclose(m.variance, var)
    assert m.event_shape == torch.Size([3, 2]) and m.batch_shape == torch.Size([2])
    assert torch.allclose(m.entropy(), 11.80326 * torch.ones(2, dtype=dt))
    assert torch.allclose(m.log_prob(torch.zeros(2, 3, 2, dtype=dt)), -14.52826 * torch.ones(2, dtype=dt))
    assert torch.allclose(m.log_prob(torch.zeros(3, 2, 3, 2, dtype=dt)), -14.52826 * torch.ones(3, 2, dtype=dt))
    assert m.sample(torch.Size([3, 4])).shape == (3, 4, 2, 3, 2)


def test_log_prob():
    torch.manual_seed(0)
    mean, var = torch.randn(4, 3), torch.randn(12).abs_()
    values = mean + 0.5
    diffs = (values - mean).view(-1)
    res = MT(mean, DiagLinearOperator(var)).log_prob(values)
    assert abs((res - (-0.5 * (math.log(math.pi * 2) * 12 + var.log().sum() + (diffs / var * diffs).sum()))) / res) < 1e-2
    mean, var = torch.randn(3, 4, 3), torch.randn(3, 12).abs_()
    values = mean + 0.5
    diffs = (values - mean).view(3, -1)
    res = MT(mean, DiagLinearOperator(var)).log_prob(values)
    assert ((res - (-0.5 * (math.log(math.pi * 2) * 12 + var.log().sum(-1) + (diffs / var * diffs).sum(-1)))) / res).abs().norm() < 1e-2


@pytest.mark.parametrize("interleaved", [True, False])
def test_to_data_independent_dist(interleaved):
    torch.manual_seed(0)
    factor = torch.randn(4, 4)
    data_covar = factor.mT @ factor
    task_covar = torch.tensor([[1.0, 0.3, 0.1], [0.3, 1.0, 0.3], [0.1, 0.3, 1.0]])
    covar = torch.kron(data_covar, task_covar) if interleaved else torch.kron(task_covar, data_covar)
    mean = torch.randn(4, 3)
    res = MT(mean, covar, interleaved=interleaved).to_data_independent_dist(jitter_val=1e-4)
    assert torch.equal(res.mean, mean)
    torch.testing.assert_close(res.covariance_matrix, data_covar.diagonal().view(-1, 1, 1) * task_covar + torch.eye(3) * 1e-4)


def test_from_batch_and_repeated_mvn():
    torch.manual_seed(0)
    mean, var = torch.randn(2, 3), torch.randn(2, 3).clamp_min(1e-6)
    mm = MT.from_batch_mvn(MVN(mean, DiagLinearOperator(var)), task_dim=-1)
    assert isinstance(mm, MT) and mm.batch_shape == torch.Size([]) and mm.event_shape == torch.Size([3, 2]) and mm.covariance_matrix.shape == (6, 6)
    assert torch.equal(mm.mean, mean.mT) and torch.allclose(mm.variance, var.mT)
    mean, var = torch.randn(2, 4, 3), torch.randn(2, 4, 3).clamp_min(1e-6)
    mm = MT.from_batch_mvn(MVN(mean, DiagLinearOperator(var)), task_dim=0)
    assert mm.batch_shape == torch.Size([4]) and mm.event_shape == torch.Size([3, 2]) and mm.covariance_matrix.shape == (4, 6, 6)
    assert torch.equal(mm.mean, mean.permute(1, 2, 0)) and torch.allclose(mm.variance, var.permute(1, 2, 0))
    # full member covariances land on the interleaved block diagonal
    a = torch.randn(2, 3, 3)
    k = a @ a.mT + torch.eye(3)
    mm = MT.from_batch_mvn(MVN(torch.zeros(2, 3), k))
    dense = mm.covariance_matrix.view(3, 2, 3, 2)
    assert torch.allclose(dense[:, 0, :, 0], k[0]) and torch.allclose(dense[:, 1, :, 1], k[1]) and float(dense[:, 0, :, 1].abs().max()) == 0.0
    mean, var = torch.randn(2, 3), torch.randn(2, 3).clamp_min(1e-6)
    mm = MT.from_repeated_mvn(MVN(mean, DiagLinearOperator(var)), num_tasks=4)
    assert mm.batch_shape == torch.Size([2]) and mm.event_shape == torch.Size([3, 4]) and mm.covariance_matrix.shape == (2, 12, 12)
    for i in range(4):
        assert torch.equal(mm.mean[..., i], mean) and torch.allclose(mm.variance[..., i], var)


@pytest.mark.parametrize("dt", [torch.float, torch.double])
def test_from_independent_mvns(dt):
    torch.manual_seed(0)
    n_tasks, n = 2, 4
    mvns = [MVN(torch.randn(4, dtype=dt), DiagLinearOperator(torch.randn(n, dtype=dt).abs_())) for _ in range(n_tasks)]
    mm = MT.from_independent_mvns(mvns)
    assert list(mm.mean.shape) == [n, n_tasks] and list(mm.covariance_matrix.shape) == [n * n_tasks] * 2
    for t in range(n_tasks):
        assert torch.equal(mm.mean[:, t], mvns[t].mean) and torch.allclose(mm.variance[:, t], mvns[t].variance)
    mvns[1] = mvns[1].expand(torch.Size([3]))                         # mixed batch shapes: the others are expanded to match
    expected = mm.expand(torch.Size([3]))
    mm = MT.from_independent_mvns(mvns)
    assert torch.equal(mm.mean, expected.mean) and torch.equal(mm.covariance_matrix, expected.covariance_matrix)
    mvns = [MVN(torch.randn(3, n, dtype=dt), DiagLinearOperator(torch.randn(3, n, dtype=dt).abs_())) for _ in range(n_tasks)]
    mm = MT.from_independent_mvns(mvns)
    assert list(mm.mean.shape) == [3, n, n_tasks] and list(mm.covariance_matrix.shape) == [3, n * n_tasks, n * n_tasks]
    with pytest.raises(ValueError):
        MT.from_independent_mvns(mvns[:1])
