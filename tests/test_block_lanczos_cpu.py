"""Host logic of the multi-vector Lanczos interface and of the block Lanczos behind the LOVE cache, on CPU in float64 with a dense
operator (no GPU, no HIP library): the product's torch formulation against the oracle's restatement of the reference's
``lanczos_tridiag(init_vecs=[n, b])`` / ``_postprocess_lanczos_root_inv_decomp`` (gpytorch/__init__.py:190-216), the block form
against its textbook properties, and the claim the block form rests on -- the variance error of a LOVE cache follows the RANK of the
Krylov space, not the number of start vectors that generated it."""
import pytest
import torch

from gpytorch_amd import lanczos as LZ
from gpytorch_amd import settings
from oracle import lanczos as OL

DT = torch.float64


def _problem(n=600, d=3, ell=0.25, noise=0.1, seed=0):
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(n, d, dtype=DT, generator=g)
    K = torch.exp(-0.5 * torch.cdist(X, X) ** 2 / ell**2) + noise * torch.eye(n, dtype=DT)
    return X, K


def _mv(K):
    return lambda q_rows: q_rows @ K        # probe-major rows: (K q^T)^T = q K for symmetric K


def test_lock_step_recurrences_equal_the_oracle_chain_by_chain():
    n, b, k = 400, 5, 30
    _, K = _problem(n)
    g = torch.Generator().manual_seed(1)
    init = torch.randn(n, b, dtype=DT, generator=g)
    Q, T = LZ.lanczos_tridiag_multi(_mv(K), n, k, init.t().contiguous())
    Qo, To = OL.lanczos_tridiag_batch(lambda v: K @ v, k, n, init)
    assert Q.shape == (b, k, n) and T.shape == (b, k, k)
    assert torch.allclose(T, To, atol=1e-9)
    assert torch.allclose(Q, Qo.transpose(1, 2), atol=1e-8)
    for i in range(b):      # ... and every chain is the single-vector recurrence of its own start vector
        Q1, T1 = OL.lanczos_tridiag(lambda v: K @ v, k, n, init[:, i : i + 1])
        assert torch.allclose(T[i], T1, atol=1e-8)
        assert torch.allclose(Q[i].t(), Q1, atol=1e-7)


def test_selection_by_test_vectors_is_the_reference_rule():
    n, b, k, c = 400, 4, 12, 3
    X, K = _problem(n)
    g = torch.Generator().manual_seed(2)
    init = torch.randn(n, b, dtype=DT, generator=g)
    test = torch.exp(-0.5 * torch.cdist(X, torch.rand(c, 3, dtype=DT, generator=g)) ** 2 / 0.25**2)
    with settings.tridiagonal_jitter(0.0):       # (the oracle's root carries no jitter)
        rt = LZ.root_inv_decomposition(None, None, None, max_iter=k, init_vec_t=init.t().contiguous(), test_vec_t=test.t().contiguous(),
                                       matvec=_mv(K), nvec=n, device=torch.device("cpu"))
    Ro, idx = OL.root_inv_decomposition_multi(lambda v: K @ v, n, k, init, test)
    assert torch.allclose(rt.t() @ rt, Ro @ Ro.t(), atol=1e-8)
    # the winner really has the smallest summed residual, by brute force over every start vector
    sums = []
    for i in range(b):
        Ri = OL.root_inv_decomposition(lambda v: K @ v, n, k, init[:, i : i + 1])
        sums.append(float((K @ (Ri @ (Ri.t() @ test)) - test).norm(dim=0).sum()))
    assert idx == min(range(b), key=sums.__getitem__)


def test_several_start_vectors_without_test_vectors_is_an_error():
    n = 64
    _, K = _problem(n)
    with pytest.raises(ValueError, match="test_vectors"):
        LZ.root_inv_decomposition(None, None, None, max_iter=8, init_vec_t=torch.randn(3, n, dtype=DT), matvec=_mv(K), nvec=n,
                                  device=torch.device("cpu"))


@pytest.mark.parametrize("b,steps", [(1, 24), (4, 10), (8, 6)])
def test_block_lanczos_properties(b, steps):
    n = 500
    _, K = _problem(n)
    g = torch.Generator().manual_seed(3)
    init = torch.randn(b, n, dtype=DT, generator=g)
    Q, T = LZ.block_lanczos(_mv(K), n, torch.device("cpu"), steps, init)
    m = steps * b
    assert Q.shape == (m, n) and T.shape == (m, m)
    assert torch.allclose(Q @ Q.t(), torch.eye(m, dtype=DT), atol=1e-10)
    assert torch.allclose(T, Q @ K @ Q.t(), atol=1e-8)
    # block tridiagonal: nothing beyond the first block off-diagonal
    blk = (torch.arange(m) // b)
    far = (blk.unsqueeze(0) - blk.unsqueeze(1)).abs() > 1
    assert T[far].abs().max() < 1e-7 if far.any() else True
    # same space as the oracle's block Lanczos: equal projectors
    Qo, To = OL.block_lanczos(lambda v: K @ v, steps, n, init.t().contiguous())
    assert torch.allclose(Q.t() @ Q, Qo @ Qo.t(), atol=1e-7)
    assert torch.allclose(torch.linalg.eigvalsh(T), torch.linalg.eigvalsh(To), atol=1e-7)
    if b == 1:      # one row per block: the ordinary Lanczos tridiagonal (up to the signs of the basis vectors)
        Q1, T1 = OL.lanczos_tridiag(lambda v: K @ v, steps, n, init.t().contiguous())
        assert torch.allclose(T.abs(), T1.abs(), atol=1e-7)


def test_a_rank_deficient_block_cuts_the_decomposition():
    n = 60
    # A has only 10 distinct eigen-directions above a flat floor: the block Krylov space of 4 vectors is exhausted after a few steps
    g = torch.Generator().manual_seed(4)
    U, _ = torch.linalg.qr(torch.randn(n, n, dtype=DT, generator=g))
    lam = torch.ones(n, dtype=DT)
    lam[:6] = torch.arange(2, 8, dtype=DT)
    K = (U * lam) @ U.t()
    init = U[:, :6].t()[:4].contiguous() + 0.0           # start inside a 6-dimensional invariant subspace
    Q, T = LZ.block_lanczos(_mv(K), n, torch.device("cpu"), 5, init)
    assert Q.shape[0] < 20 and Q.shape[0] % 4 == 0
    assert torch.allclose(Q @ Q.t(), torch.eye(Q.shape[0], dtype=DT), atol=1e-8)


def test_love_variance_error_follows_the_rank_not_the_block_size():
    n = 3000
    X, K = _problem(n, seed=5)
    g = torch.Generator().manual_seed(6)
    Xs = torch.rand(200, 3, dtype=DT, generator=g)
    Ks = torch.exp(-0.5 * torch.cdist(X, Xs) ** 2 / 0.25**2)
    exact = (Ks * torch.cholesky_solve(Ks, torch.linalg.cholesky(K))).sum(0)

    def err(root_t):
        return float(((root_t @ Ks) ** 2).sum(0).sub(exact).abs().max() / 0.1)          # in units of the noise

    dev = torch.device("cpu")
    errs = {}
    for rank in (96, 192):
        for b in (1, 8):
            with settings.lanczos_block_size(b):
                rt = LZ.root_inv_decomposition(None, None, None, max_iter=rank, matvec=_mv(K), nvec=n, device=dev, generator=g, dtype=DT,
                                               init_vec_t=None if b > 1 else torch.randn(1, n, dtype=DT, generator=g))
            assert rt.shape[0] == rank
            errs[rank, b] = err(rt)
    # same rank -> same error to within a factor two either way; doubling the rank gains orders of magnitude with either generator
    for rank in (96, 192):
        assert 0.5 < errs[rank, 8] / errs[rank, 1] < 2.0, errs
    assert errs[192, 8] < 0.05 * errs[96, 8] and errs[192, 1] < 0.05 * errs[96, 1], errs
    assert errs[192, 8] < 0.05, errs            # the reference's criterion (test_simple_gp_regression.py:436-442) at this size


def test_block_size_setting():
    # (the reference-default rank 100 keeps the reference's single-vector recurrence: a block cache of that rank is 1.3-1.6 x less accurate)
    # (round 6: 32 rows per product from 262 144 points on -- up to 32 columns cost one generation-bound launch there)
    assert LZ.block_size_for(500_000, 400) == 32 and LZ.block_size_for(500_000, 200) == 32 and LZ.block_size_for(500_000, 100) == 1
    assert LZ.block_size_for(100_000, 400) == 8 and LZ.block_size_for(100_000, 200) == 8 and LZ.block_size_for(100_000, 100) == 1
    assert LZ.block_size_for(2000, 400) == 1 and LZ.block_size_for(500_000, 16) == 1
    with settings.lanczos_block_size(1):
        assert LZ.block_size_for(500_000, 400) == 1
    with settings.lanczos_block_size(16):
        assert LZ.block_size_for(300, 100) == 16 and LZ.block_size_for(300, 10) == 10
