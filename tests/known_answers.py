"""Known answers the reference's OWN unit tests hold for composed kernels (``test/kernels/test_additive_and_product_kernels.py:33-157``: the inputs, the
closed forms and the four-digit literals are that file's data), run through this repository's kernel API.  Shared by the CPU wiring test (native entry
points doubled) and the device test."""
import torch


def check_sum_product_known_answers(g, dev):
    RBF = g.kernels.RBFKernel
    a = torch.tensor([4.0, 2.0, 8.0], device=dev).view(3, 1)
    b = torch.tensor([0.0, 2.0], device=dev).view(2, 1)
    b3 = torch.tensor([0.0, 2.0, 2.0], device=dev).view(3, 1)
    base = torch.tensor([[16.0, 4.0], [4.0, 0.0], [64.0, 36.0]], device=dev).mul(-0.5).div(2.0 ** 2).exp()

    def k(**kw):
        return RBF(**kw).initialize(lengthscale=2.0).to(dev)

    out = {}
    for name, kern, want in (
        ("product", k() * k(), base ** 2),                                                          # :33-46
        ("product_batch", k(batch_shape=torch.Size([4])) * k(), (base ** 2).repeat(4, 1, 1)),       # :48-62
        ("sum", k() + k(), base * 2),                                                               # :64-77
        ("sum_batch", k(batch_shape=torch.Size([4])) + k(), (base * 2).repeat(4, 1, 1)),            # :79-90 (the batch twin)
    ):
        kern.eval()
        with torch.no_grad():
            res = kern(a, b).to_dense()
        assert res.shape == want.shape, (name, res.shape)
        out[name] = float((res - want).norm())
        assert out[name] < 2e-5, (name, out[name])
    for name, kern, want in (
        ("sum_diag", k() + k(), [0.2702, 2.000, 0.0222]),                                           # :92-107
        ("sum_of_three_diag", k() + k() + k(), [0.4060, 3.000, 0.0333]),                            # :109-124
        ("product_of_three_diag", k() * k() * k(), [2.4788e-03, 1.000, 1.3710e-06]),                # :126-142
        ("product_diag", k() * k(), [1.8316e-02, 1.000, 1.2341e-04]),                               # :144-157
    ):
        kern.eval()
        with torch.no_grad():
            res = kern(a, b3, diag=True)
        out[name] = float((res - torch.tensor(want, device=dev)).norm())
        assert out[name] < 1e-3, (name, res.tolist(), want)
    # the same inputs on both sides: the diagonal of K(x, x) is the number of summands / one
    with torch.no_grad():
        assert torch.allclose((k() + k())(a, a, diag=True), torch.full((3,), 2.0, device=dev))
        assert torch.allclose((k() * k())(a, diag=True), torch.ones(3, device=dev))
        assert torch.allclose((k() * k())(a, a.clone(), diag=True), torch.ones(3, device=dev))
    return out


def check_stationary_and_periodic_unit_tests(g, dev):
    """The closed forms of ``test/kernels/test_rbf_kernel.py:20-125`` (ARD, ARD in a batch, separate lengthscales per batch member, active dimensions,
    ``last_dim_is_batch`` with ARD lengthscales -- each input dimension scaled by ITS lengthscale before it becomes a batch member) and of
    ``test/kernels/test_periodic_kernel.py:20-88`` (the periodic function, a [1, 1, 1]-shaped hyper-parameter given to a kernel without a batch shape,
    separate periods per batch member)."""
    import math

    RBF, Periodic = g.kernels.RBFKernel, g.kernels.PeriodicKernel
    T = lambda *a, **k: torch.tensor(*a, dtype=torch.float, device=dev, **k)  # noqa: E731
    dn = lambda x: float(x.norm())  # noqa: E731
    with torch.no_grad():
        # test_ard
        a, b, ls = T([[1, 2], [2, 4]]), T([[1, 3], [0, 4]]), T([1, 2]).view(1, 2)
        k = RBF(ard_num_dims=2).to(dev)
        k.initialize(lengthscale=ls)
        k.eval()
        sa, sb = a / ls, b / ls
        actual = (sa.unsqueeze(-2) - sb.unsqueeze(-3)).pow(2).sum(-1).mul(-0.5).exp()
        assert dn(k(a, b).to_dense() - actual) < 1e-5
        assert dn(k(a, b).diagonal(dim1=-1, dim2=-2) - actual.diagonal()) < 1e-5
        per_dim = (sa.mT.unsqueeze(-1) - sb.mT.unsqueeze(-2)).pow(2).mul(-0.5).exp()
        res = k(a, b, last_dim_is_batch=True)
        assert dn(res.to_dense() - per_dim) < 1e-5
        assert dn(res.diagonal(dim1=-1, dim2=-2) - per_dim.diagonal(dim1=-1, dim2=-2)) < 1e-5
        # test_ard_batch / test_ard_separate_batch
        a = T([[[1, 2, 3], [2, 4, 0]], [[-1, 1, 2], [2, 1, 4]]])
        b = T([[[1, 3, 1]], [[2, -1, 0]]]).repeat(1, 2, 1)
        for ls in (T([[[1, 2, 1]]]), T([[[1, 2, 1]], [[2, 1, 0.5]]])):
            k = RBF(batch_shape=torch.Size([2]), ard_num_dims=3).to(dev)
            k.initialize(lengthscale=ls)
            k.eval()
            sa, sb = a / ls, b / ls
            actual = (sa.unsqueeze(-2) - sb.unsqueeze(-3)).pow(2).sum(-1).mul(-0.5).exp()
            assert dn(k(a, b).to_dense() - actual) < 1e-5
            assert dn(k(a, b).diagonal(dim1=-1, dim2=-2) - actual.diagonal(dim1=-1, dim2=-2)) < 1e-5
            per_dim = (sa.mT.unsqueeze(-1) - sb.mT.unsqueeze(-2)).pow(2).mul(-0.5).exp()
            res = k(a, b, last_dim_is_batch=True)
            assert res.shape == per_dim.shape and dn(res.to_dense() - per_dim) < 1e-5
        # test_subset_active_compute_radial_basis_function
        a = torch.cat((T([4, 2, 8]).view(3, 1), T([1, 2, 3]).view(3, 1)), 1)
        b = T([0, 2, 4]).view(3, 1)
        k = RBF(active_dims=[0]).to(dev)
        k.initialize(lengthscale=2)
        k.eval()
        actual = T([[16, 4, 0], [4, 0, 4], [64, 36, 16]]).mul(-0.5).div(4).exp()
        assert dn(k(a, b).to_dense() - actual) < 1e-5
        # periodic: test_computes_periodic_function, test_batch, test_batch_separate
        a, b = T([4, 2, 8]).view(3, 1), T([0, 2]).view(2, 1)
        k = Periodic().initialize(lengthscale=2, period_length=3).to(dev)
        k.eval()
        actual = torch.exp(-2 * torch.sin(math.pi * (a - b.t()) / 3).pow(2) / 2)
        assert dn(k(a, b).to_dense() - actual) < 1e-5
        b3 = T([0, 2, 2]).view(3, 1)
        assert dn(k(a, b3, diag=True) - k(a, b3).to_dense().diagonal()) < 1e-6        # (two different inputs: the elementwise diagonal)
        a, b = T([[4, 2, 8], [1, 2, 3]]).view(2, 3, 1), T([[0, 2], [-1, 2]]).view(2, 2, 1)
        k = Periodic().initialize(lengthscale=T(2).view(1, 1, 1), period_length=T(1).view(1, 1, 1)).to(dev)
        k.eval()
        assert dn(k(a, b).to_dense() - torch.stack([k(a[i], b[i]).to_dense() for i in range(2)])) < 1e-5
        period, ls = T([1, 2]).view(2, 1, 1), T([2, 1]).view(2, 1, 1)
        k = Periodic(batch_shape=torch.Size([2])).initialize(lengthscale=ls, period_length=period).to(dev)
        k.eval()
        actual = torch.stack([((a[i].unsqueeze(1) - b[i].unsqueeze(0)) * math.pi / period[i].unsqueeze(-1)).sin().pow(2).sum(-1).div(ls[i]).mul(-2.0).exp() for i in range(2)])
        assert dn(k(a, b).to_dense() - actual) < 1e-5


def check_scale_kernel_unit_tests(g, dev):
    """``test/kernels/test_scale_kernel.py:24-127``: an outputscale over an ARD kernel (dense, diagonal, ``last_dim_is_batch``), a per-member outputscale
    over a batch kernel (the outputscale follows the kernel's batch dimensions, NOT the dimension-batch ``last_dim_is_batch`` appends), ``initialize``,
    stationarity, inherited active dimensions."""
    K = g.kernels
    T = lambda *a: torch.tensor(*a, dtype=torch.float, device=dev)  # noqa: E731
    dn = lambda x: float(x.norm())  # noqa: E731
    with torch.no_grad():
        a, b, ls = T([[1, 2], [2, 4]]), T([[1, 3], [0, 4]]), T([1, 2]).view(1, 2)
        base = K.RBFKernel(ard_num_dims=2).to(dev)
        base.initialize(lengthscale=ls)
        k = K.ScaleKernel(base).to(dev)
        k.initialize(outputscale=T([3]))
        k.eval()
        sa, sb = a / ls, b / ls
        actual = (sa.unsqueeze(-2) - sb.unsqueeze(-3)).pow(2).sum(-1).mul(-0.5).exp() * 3
        assert dn(k(a, b).to_dense() - actual) < 1e-5 and dn(k(a, b).diagonal(dim1=-1, dim2=-2) - actual.diagonal()) < 1e-5
        per_dim = (sa.mT.unsqueeze(-1) - sb.mT.unsqueeze(-2)).pow(2).mul(-0.5).exp() * 3
        res = k(a, b, last_dim_is_batch=True)
        assert dn(res.to_dense() - per_dim) < 1e-5 and dn(res.diagonal(dim1=-1, dim2=-2) - per_dim.diagonal(dim1=-1, dim2=-2)) < 1e-5
        a = T([[[1, 2, 3], [2, 4, 0]], [[-1, 1, 2], [2, 1, 4]]])
        b = T([[[1, 3, 1]], [[2, -1, 0]]]).repeat(1, 2, 1)
        ls = T([[[1, 2, 1]]])
        base = K.RBFKernel(batch_shape=torch.Size([2]), ard_num_dims=3).to(dev)
        base.initialize(lengthscale=ls)
        k = K.ScaleKernel(base, batch_shape=torch.Size([2])).to(dev)
        k.initialize(outputscale=T([1, 2]))
        k.eval()
        sa, sb = a / ls, b / ls
        actual = (sa.unsqueeze(-2) - sb.unsqueeze(-3)).pow(2).sum(-1).mul(-0.5).exp()
        actual[1] *= 2
        assert dn(k(a, b).to_dense() - actual) < 1e-5 and dn(k(a, b).diagonal(dim1=-1, dim2=-2) - actual.diagonal(dim1=-1, dim2=-2)) < 1e-5
        per_dim = (sa.mT.unsqueeze(-1) - sb.mT.unsqueeze(-2)).pow(2).mul(-0.5).exp()
        per_dim[1] *= 2
        res = k(a, b, last_dim_is_batch=True)
        assert dn(res.to_dense() - per_dim) < 1e-5 and dn(res.diagonal(dim1=-1, dim2=-2) - per_dim.diagonal(dim1=-2, dim2=-1)) < 1e-5
    k = K.ScaleKernel(K.RBFKernel())
    k.initialize(outputscale=3.14)
    assert dn(k.outputscale - torch.tensor(3.14).view_as(k.outputscale)) < 1e-5
    k = K.ScaleKernel(K.RBFKernel(), batch_shape=torch.Size([2]))
    v = torch.tensor([3.14, 4.13])
    k.initialize(outputscale=v)
    assert dn(k.outputscale - v.view_as(k.outputscale)) < 1e-5
    assert K.ScaleKernel(K.RBFKernel()).is_stationary
    base = K.RBFKernel(active_dims=(1, 2), ard_num_dims=2)
    assert torch.all(K.ScaleKernel(base).active_dims == base.active_dims)
