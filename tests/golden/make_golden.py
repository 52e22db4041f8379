#!/usr/bin/env python3
"""Generate golden fixtures by EXECUTING the reference's own code in this container.

Run once here (``python tests/golden/make_golden.py``); the .npz files it writes are committed
and are what travels to the GPU box (``/root/reference`` does not exist there).

What is executed from /root/reference (nothing is copied into the repo):
  * ``gpytorch/functions/rbf_covariance.py``     -> RBFCovariance.apply   (imports only torch)
  * ``gpytorch/functions/matern_covariance.py``  -> MaternCovariance.apply (imports only torch, math)
  * ``gpytorch/utils/transforms.py``             -> inv_softplus, inv_sigmoid (imports only torch)
  * ``gpytorch/kernels/kernel.py`` lines 26-60   -> sq_dist, dist: the two function definitions are
    extracted with ``ast`` and exec'd (the module itself cannot be imported because the third-party
    ``linear_operator`` package is not installed in this image).
  * ``gpytorch/kernels/kernel.py:307-352``       -> Kernel.covar_dist, ``gpytorch/kernels/periodic_kernel.py:125-142`` ->
    PeriodicKernel.forward, ``gpytorch/kernels/rq_kernel.py:61-74`` -> RQKernel.forward: the three METHOD definitions are
    extracted with ``ast`` and bound to a stub object that only carries the hyper-parameter tensors in the shapes the real
    modules hold them (lengthscale / period_length [1, d or 1], alpha [1]) -> ``composite_values.npz``.

The BBMM arithmetic itself lives in ``linear_operator`` (absent) so no golden vectors for CG /
Lanczos / pivoted Cholesky can be generated: those stay pinned to dense float64 Cholesky.
"""
from __future__ import annotations

import ast
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference/gpytorch"
OUT = os.path.dirname(os.path.abspath(__file__))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _extract_functions(path, names):
    src = open(path).read()
    tree = ast.parse(src)
    ns = {"torch": torch}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            code = compile(ast.Module(body=[node], type_ignores=[]), path, "exec")
            exec(code, ns)
    return [ns[n] for n in names]


def _extract_method(path, cls, name, ns):
    """The FunctionDef ``name`` of class ``cls`` in ``path``, compiled in namespace ``ns`` (annotations are dropped: they name
    typing aliases the namespace does not need)."""
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name == name:
                    item.returns = None
                    for a in item.args.args + item.args.kwonlyargs:
                        a.annotation = None
                    item.decorator_list = []
                    exec(compile(ast.Module(body=[item], type_ignores=[]), path, "exec"), ns)
                    return ns[name]
    raise KeyError((cls, name))


def composite_fixtures(sq_dist, dist):
    """Outputs of the reference's own PeriodicKernel.forward / RQKernel.forward (through its own Kernel.covar_dist)."""
    import math

    ns = {"torch": torch, "math": math, "sq_dist": sq_dist, "dist": dist}
    covar_dist = _extract_method(f"{REF}/kernels/kernel.py", "Kernel", "covar_dist", dict(ns))
    per_fwd = _extract_method(f"{REF}/kernels/periodic_kernel.py", "PeriodicKernel", "forward", dict(ns))
    rq_fwd = _extract_method(f"{REF}/kernels/rq_kernel.py", "RQKernel", "forward", dict(ns))
    Per = type("RefPeriodic", (), {"covar_dist": covar_dist, "forward": per_fwd})
    Rq = type("RefRQ", (), {"covar_dist": covar_dist, "forward": rq_fwd})
    out = {}
    cases = [  # name, n, m, d, ard, same, dtype
        ("p", 41, 41, 1, False, True, torch.float64),
        ("q", 35, 52, 3, False, False, torch.float64),
        ("r", 48, 48, 3, True, True, torch.float64),
        ("s", 30, 44, 2, True, False, torch.float32),
    ]
    for name, n, m, d, ard, same, dt in cases:
        g = torch.Generator().manual_seed(1000 + ord(name))
        x1 = torch.rand(n, d, generator=g, dtype=dt) * 3
        x2 = x1.clone() if same else torch.rand(m, d, generator=g, dtype=dt) * 3
        k = d if ard else 1
        ls = (0.5 + 1.5 * torch.rand(1, k, generator=g, dtype=dt))
        period = (0.6 + 1.4 * torch.rand(1, k, generator=g, dtype=dt))
        alpha = (0.4 + 2.0 * torch.rand(1, generator=g, dtype=dt))
        W = torch.randn(n, x2.shape[0], generator=g, dtype=dt)
        out.update({f"{name}_x1": x1.numpy(), f"{name}_x2": x2.numpy(), f"{name}_ls": ls.numpy(), f"{name}_period": period.numpy(),
                    f"{name}_alpha": alpha.numpy(), f"{name}_W": W.numpy(), f"{name}_same": np.array(same)})
        # PeriodicKernel.forward (periodic_kernel.py:125-142), value + d sum(W K) / d(lengthscale, period_length)
        pk = Per()
        pk.lengthscale = ls.clone().requires_grad_(True)
        pk.period_length = period.clone().requires_grad_(True)
        kp = pk.forward(x1, x2)
        gl, gp = torch.autograd.grad((kp * W).sum(), [pk.lengthscale, pk.period_length])
        out[f"{name}_periodic"] = kp.detach().numpy()
        out[f"{name}_periodic_dls"] = gl.numpy()
        out[f"{name}_periodic_dperiod"] = gp.numpy()
        out[f"{name}_periodic_diag"] = pk.forward(x1, x1, diag=True).detach().numpy()
        # RQKernel.forward (rq_kernel.py:61-74), value + d sum(W K) / d(lengthscale, alpha)
        rk = Rq()
        rk.lengthscale = ls.clone().requires_grad_(True)
        rk.alpha = alpha.clone().requires_grad_(True)
        kr = rk.forward(x1, x2)
        gl, ga = torch.autograd.grad((kr * W).sum(), [rk.lengthscale, rk.alpha])
        out[f"{name}_rq"] = kr.detach().numpy()
        out[f"{name}_rq_dls"] = gl.numpy()
        out[f"{name}_rq_dalpha"] = ga.numpy()
        out[f"{name}_rq_diag"] = rk.forward(x1, x1, diag=True).detach().numpy()
    np.savez_compressed(os.path.join(OUT, "composite_values.npz"), **out)
    print("wrote composite_values.npz with", len(out), "arrays")


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not mounted; fixtures are generated in the build container only")
    rbf_mod = _load(f"{REF}/functions/rbf_covariance.py", "ref_rbf_covariance")
    mat_mod = _load(f"{REF}/functions/matern_covariance.py", "ref_matern_covariance")
    sq_dist, dist = _extract_functions(f"{REF}/kernels/kernel.py", ["sq_dist", "dist"])

    out = {}
    cases = [  # (name, n, m, d, lengthscale, dtype, same)
        ("a", 37, 37, 3, 0.25, torch.float64, True),
        ("b", 64, 45, 3, 0.6, torch.float64, False),
        ("c", 50, 50, 10, 0.8, torch.float64, True),
        ("d", 33, 70, 6, 0.5, torch.float32, False),
        ("e", 40, 40, 1, 2.0, torch.float32, True),
    ]
    for name, n, m, d, ls, dt, same in cases:
        g = torch.Generator().manual_seed(ord(name))
        x1 = torch.rand(n, d, generator=g, dtype=dt)
        x2 = x1.clone() if same else torch.rand(m, d, generator=g, dtype=dt)
        lsz = torch.tensor([[ls]], dtype=dt, requires_grad=True)
        out[f"{name}_x1"] = x1.numpy()
        out[f"{name}_x2"] = x2.numpy()
        out[f"{name}_ls"] = np.array(ls)
        out[f"{name}_same"] = np.array(same)
        x1_eq_x2 = torch.equal(x1, x2)
        out[f"{name}_sq_dist"] = sq_dist(x1, x2, x1_eq_x2).numpy()
        out[f"{name}_dist"] = dist(x1, x2, x1_eq_x2).numpy()
        # RBF: value + lengthscale gradient of sum(W * K) for a fixed W (exercises backward :26-29)
        W = torch.randn(n, x2.shape[0], generator=g, dtype=dt)
        out[f"{name}_W"] = W.numpy()
        k = rbf_mod.RBFCovariance.apply(x1, x2, lsz, lambda a, b: sq_dist(a, b, x1_eq_x2))
        (gl,) = torch.autograd.grad((k * W).sum(), lsz)
        out[f"{name}_rbf"] = k.detach().numpy()
        out[f"{name}_rbf_dls"] = gl.numpy()
        for nu in (0.5, 1.5, 2.5):
            k = mat_mod.MaternCovariance.apply(x1, x2, lsz, nu, lambda a, b: dist(a, b, x1_eq_x2))
            (gl,) = torch.autograd.grad((k * W).sum(), lsz)
            out[f"{name}_matern{int(nu * 10):02d}"] = k.detach().numpy()
            out[f"{name}_matern{int(nu * 10):02d}_dls"] = gl.numpy()
    np.savez_compressed(os.path.join(OUT, "kernel_values.npz"), **out)
    print("wrote kernel_values.npz with", len(out), "arrays")
    composite_fixtures(sq_dist, dist)

    # parameter transforms (gpytorch/utils/transforms.py imports only torch): the raw <-> constrained maps every
    # hyper-parameter of the path goes through (constraints/constraints.py:172-194 call these)
    tr = _load(f"{REF}/utils/transforms.py", "ref_transforms")
    xs = torch.cat([torch.logspace(-6, 2, 50, dtype=torch.float64), torch.tensor([1e-4, 0.1, 0.6931, 1.0, 20.0], dtype=torch.float64)])
    ps = torch.linspace(0.001, 0.999, 41, dtype=torch.float64)
    tout = {"x": xs.numpy(), "inv_softplus": tr.inv_softplus(xs).numpy(), "p": ps.numpy(), "inv_sigmoid": tr.inv_sigmoid(ps).numpy()}
    np.savez_compressed(os.path.join(OUT, "transform_values.npz"), **tout)
    print("wrote transform_values.npz")


if __name__ == "__main__":
    main()
