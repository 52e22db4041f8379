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
