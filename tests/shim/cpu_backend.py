"""TEST-ONLY CPU double of the few ``gpytorch_amd.backend`` entry points the SMALL-n (dense Cholesky) branches of the operators use, so that the
host-side wiring -- the reference's own Python layers driving the operator protocol (tests/test_reference_layers_cpu.py) -- can execute in
a container without a GPU.  It restates the prepared-point convention of ``gpamd_prep_points_f32`` (z = coef (x - shift) / l,
k = exp2(-|dz|^2) for RBF, poly_nu(r) exp(-r) with r = |dz| for Matern) in float64 torch; the tests that use it check it against
``oracle/`` (the reference's formulas) first.  The product never imports this file and has no CPU path of its own: without the patch
every one of these functions raises on a CPU tensor.  The BBMM branches (mBCG, Lanczos, pivoted Cholesky: native kernels only) are NOT
doubled -- they are covered on the device by the ``-m gpu`` suite.
"""
import math

import torch

from gpytorch_amd import backend as B

LN2 = math.log(2.0)


def prep_points(kind, x, lengthscale, shift=None, param=None):
    n, d = x.shape[-2], x.shape[-1]
    dp = B.padded_dim(d)
    wd = B.work_dtype(x)
    xs = x.detach().to(torch.float64)
    if shift is not None:
        xs = xs - shift.detach().to(torch.float64).reshape(-1)
    ls = lengthscale.detach().to(torch.float64).reshape(-1)
    if ls.numel() not in (1, d):
        raise ValueError(f"lengthscale must have 1 or {d} elements, got {ls.numel()}")
    z = torch.zeros(n, dp, dtype=torch.float64)
    if kind == "rq":                      # z = (x - shift) / (l sqrt(2 alpha)), k = (1 + |dz|^2)^-alpha  (backend.prep_coef_of; VALUES only: no derivative double)
        param = float(param)
        z[:, :d] = xs / ls / (2.0 * param) ** 0.5
        return B.PreparedPoints(z.to(wd), n, d, dp, kind, param)
    z[:, :d] = B.prep_coef(kind) * xs / ls
    return B.PreparedPoints(z.to(wd), n, d, dp, kind, None)


def _k_and_dk(x1, x2):
    """(k, dk/ds, per-dimension squared differences [n, m, d]) in float64, s = |z_i - z_j|^2."""
    z1, z2 = x1.xp[:, : x1.d].double(), x2.xp[:, : x2.d].double()
    sq = (z1[:, None, :] - z2[None, :, :]).pow(2)
    s = sq.sum(-1)
    kind = x1.kind
    if kind == "rq":
        k = (1.0 + s).pow(-x1.param)
        dk = -x1.param * (1.0 + s).pow(-x1.param - 1.0)
    elif kind == "rbf":
        k = torch.exp2(-s)
        dk = -LN2 * k
    else:
        r = s.sqrt()
        e = torch.exp(-r)
        if kind == "matern12":
            k = e
            dk = torch.where(r > 0, -e / (2 * r.clamp_min(1e-300)), torch.zeros_like(r))
        elif kind == "matern32":
            k = (1 + r) * e
            dk = -0.5 * e
        else:
            k = (1 + r + r * r / 3) * e
            dk = -(1 + r) * e / 6
    return k, dk, sq


def kernel_dense(x1, x2, scale=None):
    k = _k_and_dk(x1, x2)[0]
    if scale is not None:
        k = k * scale.double().reshape(())
    return k.to(x1.dtype)


def kernel_rows(x1, rows, x2, scale=None):
    sub = B.PreparedPoints(x1.xp[rows.reshape(-1).long()], rows.numel(), x1.d, x1.dp, x1.kind, x1.param)
    return kernel_dense(sub, x2, scale)


def kernel_diag(x1, x2, scale=None):
    assert x1.n == x2.n
    return kernel_dense(x1, x2, scale).diagonal().clone()


def kv(x1, x2, vt, scale=None, dscale=None, vd=None, out=None, dvec=None):
    n, m = x1.n, x2.n
    res = torch.zeros(vt.shape[0], B.round_up(n, 4), dtype=torch.float64)
    res[:, :n] = (kernel_dense(x1, x2, scale).double() @ vt[:, :m].double().t()).t()
    if vd is not None and (dscale is not None or dvec is not None):
        dtot = torch.zeros(n, dtype=torch.float64)
        if dscale is not None:
            dtot += dscale.double().reshape(())
        if dvec is not None:
            dtot += dvec[:n].double()
        res[:, :n] += vd[:, :n].double() * dtot
    res = res.to(x1.dtype)
    if out is not None:
        out.copy_(res)
        return out
    return res


def coldot(a, b, n):
    return (a[:, :n].double() * b[:, :n].double()).sum(-1).to(a.dtype)


def kv_grad2(x1, x2, lt, rt, iso=False, want_gz1=False):
    """Same return convention as ``backend.kv_grad2``: g [2 + dp] = (sum W k, per-dimension sums of W dk/ds (z_iq - z_jq)^2, ..., 0) with
    W = left^T right, and -- on request -- d/dz_i of sum_j A_ij |z_i - z_j|^2 as [n, d]."""
    k, dk, sq = _k_and_dk(x1, x2)
    W = lt[:, : x1.n].double().t() @ rt[:, : x2.n].double()
    A = W * dk
    g = torch.zeros(2 + x1.dp, dtype=torch.float64)
    g[0] = (W * k).sum()
    g[1 : 1 + x1.d] = (A.unsqueeze(-1) * sq).sum((0, 1))
    gz = None
    if want_gz1:
        z1, z2 = x1.xp[:, : x1.d].double(), x2.xp[:, : x2.d].double()
        gz = (2.0 * (z1 * A.sum(1, keepdim=True) - A @ z2)).to(x1.dtype)
    return g.to(x1.dtype), gz


def install(monkeypatch):
    """Patch the doubled entry points into ``gpytorch_amd.backend`` for the duration of one test."""
    monkeypatch.setattr(B, "_require_gpu", lambda t, name: None)
    for name, fn in (("prep_points", prep_points), ("kernel_dense", kernel_dense), ("kernel_rows", kernel_rows), ("kernel_diag", kernel_diag),
                     ("kv", kv), ("coldot", coldot), ("kv_grad2", kv_grad2)):
        monkeypatch.setattr(B, name, fn)
    monkeypatch.setattr(B, "grad_gram_ok", lambda a, b: True)
