"""The reference's generic kernel battery (``gpytorch/test/base_kernel_test_case.py:30-197``: active dimensions, batch inputs under a kernel with and
without a batch shape, ARD, ``diag=True``, kernel ``__getitem__``, pickling, dtype / device), restated against this repository's kernel classes.
Shared by the CPU wiring test (native entry points doubled) and the device test."""
import pickle

import torch


def _dense(op):
    return op.to_dense() if hasattr(op, "to_dense") else op


def run_battery(make, make_ard, dev, skip=()):
    """``make(**kwargs)`` / ``make_ard(num_dims, **kwargs)`` build the kernel (base_kernel_test_case.py:15-19).  Returns the names of the checks run."""
    gen = torch.Generator().manual_seed(0)
    x_plain = torch.randn(50, 10, generator=gen).to(dev)
    x_single = torch.randn(2, 3, 2, generator=gen).to(dev)
    x_double = torch.randn(3, 2, 50, 2, generator=gen).to(dev)
    close = lambda a, b, atol=1e-5: torch.testing.assert_close(a, b, rtol=1e-3, atol=atol)  # noqa: E731
    done = []

    def stacked(kernel, x):
        if x.dim() == 3:
            return torch.stack([_dense(kernel(x[i])) for i in range(x.shape[0])])
        return torch.stack([torch.stack([_dense(kernel(x[i, j])) for j in range(x.shape[1])]) for i in range(x.shape[0])])

    with torch.no_grad():
        for dims in ([0, 2, 4, 6], list(range(3, 9))):                                   # test_active_dims_list / _range
            close(_dense(make(active_dims=dims).to(dev)(x_plain)), _dense(make().to(dev)(x_plain[:, dims])))
        done.append("active_dims")
        for kw in ({}, {"batch_shape": torch.Size([])}):                                 # test_{no,single}_batch_kernel_single_batch_x_no_ard
            k = make(**kw).to(dev)
            actual = stacked(k, x_single)
            close(_dense(k(x_single)), actual)
            close(k(x_single, diag=True), actual.diagonal(dim1=-1, dim2=-2))
        done.append("single_batch_x")
        k = make(batch_shape=torch.Size([])).to(dev)                                     # test_no_batch_kernel_double_batch_x_no_ard
        actual = stacked(k, x_double)
        close(_dense(k(x_double)), actual, atol=5e-4)
        close(k(x_double, diag=True), actual.diagonal(dim1=-1, dim2=-2))
        done.append("double_batch_x")
        if make_ard is not None:                                                         # test_no_batch_kernel_double_batch_x_ard + the smoke twin
            k = make_ard(2, batch_shape=torch.Size([])).to(dev)
            actual = stacked(k, x_double)
            close(_dense(k(x_double)), actual)
            close(k(x_double, diag=True), actual.diagonal(dim1=-1, dim2=-2))
            k = make_ard(2, batch_shape=torch.Size([3, 2])).to(dev)
            assert _dense(k(x_double)).shape == (3, 2, 50, 50) and k(x_double, diag=True).shape == (3, 2, 50)
            done.append("ard")
        k = make(batch_shape=torch.Size([3, 2])).to(dev)                                 # test_smoke_double_batch_kernel_double_batch_x_no_ard
        assert _dense(k(x_double)).shape == (3, 2, 50, 50) and k(x_double, diag=True).shape == (3, 2, 50)
        done.append("double_batch_kernel")
        if "getitem" not in skip:
            k = make(batch_shape=torch.Size([2])).to(dev)                                # test_kernel_getitem_single_batch
            close(_dense(k(x_single))[0], _dense(k[0](x_single[0])))
            k = make(batch_shape=torch.Size([3, 2])).to(dev)                             # test_kernel_getitem_double_batch
            close(_dense(k(x_double))[0, 1], _dense(k[0, 1](x_double[0, 1])))
            k = make(batch_shape=torch.Size([2])).to(dev)                                # test_kernel_getitem_broadcast
            k = k.expand_batch(torch.broadcast_shapes(k.batch_shape, x_double.shape[:-2]))
            idx1 = torch.tensor([0, 2, 1], device=dev).unsqueeze(-1)
            idx2 = torch.tensor([1, 0, 0], device=dev).unsqueeze(-2)
            close(_dense(k(x_double))[idx1, idx2], _dense(k[idx1, idx2](x_double[idx1, idx2])), atol=5e-4)
            done.append("getitem")
    k = make(batch_shape=torch.Size([]))                                                 # test_kernel_pickle_unpickle, test_kernel_dtype_device
    pickle.loads(pickle.dumps(k))
    assert k.dtype == torch.get_default_dtype() and k.device == torch.device("cpu")
    k.to(dtype=torch.float64)
    assert k.dtype == torch.float64
    done.append("pickle_dtype")
    return done


def families(g):
    """(name, create_kernel_no_ard, create_kernel_ard) as the reference's kernel test classes define them (test/kernels/test_rbf_kernel.py:13-18,
    test_matern_kernel.py:20-37, test_periodic_kernel.py:13-18, test_rq_kernel.py:12-17, test_scale_kernel.py:13-22)."""
    K = g.kernels
    return [
        ("rbf", lambda **kw: K.RBFKernel(**kw), lambda d, **kw: K.RBFKernel(ard_num_dims=d, **kw)),
        ("matern32", lambda **kw: K.MaternKernel(nu=1.5, **kw), lambda d, **kw: K.MaternKernel(nu=1.5, ard_num_dims=d, **kw)),
        ("matern12", lambda **kw: K.MaternKernel(nu=0.5, **kw).initialize(lengthscale=5.0), lambda d, **kw: K.MaternKernel(nu=0.5, ard_num_dims=d, **kw).initialize(lengthscale=5.0)),
        ("matern52", lambda **kw: K.MaternKernel(nu=2.5, **kw), lambda d, **kw: K.MaternKernel(nu=2.5, ard_num_dims=d, **kw)),
        ("periodic", lambda **kw: K.PeriodicKernel(**kw), lambda d, **kw: K.PeriodicKernel(ard_num_dims=d, **kw)),
        ("rq", lambda **kw: K.RQKernel(**kw), lambda d, **kw: K.RQKernel(ard_num_dims=d, **kw)),
        ("scale_rbf", lambda **kw: K.ScaleKernel(K.RBFKernel(**kw), batch_shape=kw.get("batch_shape", torch.Size([]))), None),
    ]
