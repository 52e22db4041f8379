#!/usr/bin/env python3
"""Full-size runs of the BASELINE.json configurations on one MI355X (timings + size-independent checks).

  C2'  n = 500 000, d = 3, RBF, 64 probes + y     : K*V time / TFLOP/s, unit-vector check, full MLL (north-star size)
  C3   n = 500 000, d = 10, Matern-5/2, rank-100 pivoted-Cholesky preconditioner: factor time, K*V, 25 mBCG iterations
  C4/8 n = 1 000 000, d = 3, RBF, 32 probes + y   : K*V time (one GPU's share of the 8-GPU configuration)
  C5/4 n = 200 000, d = 6, T = 4 tasks, 16 probes + y: Kronecker MVM time (one GPU's share)
Writes one JSON document to stdout / the given path.
"""
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpytorch_amd import backend as B  # noqa: E402
from gpytorch_amd import settings as S  # noqa: E402
from gpytorch_amd.bbmm import LOG_2PI, build_preconditioner, inv_quad_logdet_forward  # noqa: E402
from gpytorch_amd.linear_cg import linear_cg  # noqa: E402
from gpytorch_amd.multitask import kron_matvec  # noqa: E402

dev = torch.device("cuda:0")
out = {}


def synth(n, d, seed=0):
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(n, d, generator=g)
    y = torch.sin(2 * math.pi * X[:, 0]) + torch.cos(math.pi * X.sum(-1)) + 0.1 * torch.randn(n, generator=g)
    return X.to(dev), y.to(dev)


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), r


def kv_case(name, kind, n, d, t, ls):
    X, y = synth(n, d)
    xp = B.prep_points(kind, X, torch.tensor(ls), X.mean(0))
    V = torch.randn(t, B.round_up(n, 4), device=dev)
    V[:, n:] = 0
    ms, o = timed(lambda: B.kv(xp, xp, V))
    flags = B.kv_flags(xp, xp, t)
    # unit vectors: K e_j must equal explicit row j (size-independent exactness check)
    E = torch.zeros(t, B.round_up(n, 4), device=dev)
    idx = (torch.arange(t) * (n // t) + 3).to(dev)
    E[torch.arange(t, device=dev), idx] = 1.0
    cols = B.kv(xp, xp, E)[:, :n]
    rows = B.kernel_rows(xp, idx, xp)
    err = float((cols - rows).abs().max())
    out[name] = dict(kind=kind, n=n, d=d, t=t, kv_ms=ms, tflops=2.0 * n * n * t / ms / 1e9, gram=bool(flags), zmax2=xp.zmax2,
                     unit_vector_max_abs_err=err, plan=B.kv_plan(kind, n, n, d, t, flags, B.round_up(n, 4))[:2])
    print(name, out[name], flush=True)
    return X, y, xp


# ---- C2' north-star size: K*V + full MLL
X, y, xp = kv_case("c2_500k_rbf_d3_t65", "rbf", 500_000, 3, 65, 0.25)
sc, s2 = torch.tensor([1.0], device=dev), torch.tensor([0.1], device=dev)
gen = torch.Generator(device=dev).manual_seed(1234)
torch.cuda.synchronize()
t0 = time.perf_counter()
res = inv_quad_logdet_forward(xp, sc, s2, B.to_probe_major(y.unsqueeze(-1)), num_probes=64, precond=None, generator=gen)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
n = xp.n
out["mll_500k"] = dict(seconds=dt, cg_iterations=res.info.iterations, tolerance_reached=res.info.tolerance_reached,
                       mll=float(-0.5 * (res.inv_quad.sum() + res.logdet + n * LOG_2PI) / n),
                       kv_tflops_whole_step=2.0 * n * n * 65 * res.info.iterations / dt / 1e12)
print("mll_500k", out["mll_500k"], flush=True)
del res

# ---- C3: Matern-5/2, d = 10, rank-100 preconditioner
X, y, xp = kv_case("c3_500k_matern52_d10_t65", "matern52", 500_000, 10, 65, 0.8)
torch.cuda.synchronize()
t0 = time.perf_counter()
pre = build_preconditioner(xp, sc, s2, rank=100, tol=1e-3, min_size=2000)
torch.cuda.synchronize()
out["c3_precond"] = dict(seconds=time.perf_counter() - t0, rank=int(pre.lt.shape[0]), logdet_P=float(pre.logdet))
print("c3_precond", out["c3_precond"], flush=True)
rhs = torch.zeros(65, B.round_up(n, 4), device=dev)
rhs[:64, :n] = torch.randn(64, n, device=dev, generator=gen)
rhs[64, :n] = y
import warnings  # noqa: E402

for label, p in (("precond", pre), ("noprecond", None)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sol, info = linear_cg(xp, sc, s2, rhs, n_tridiag=0, tolerance=1e-30, max_iter=25, preconditioner=p)
    torch.cuda.synchronize()
    out[f"c3_cg25_{label}"] = dict(seconds=time.perf_counter() - t0, mean_rel_residual=float(info.residual_norms.mean()))
    print(f"c3_cg25_{label}", out[f"c3_cg25_{label}"], flush=True)
del pre, sol, rhs

# ---- C4 share: n = 1e6, 32 probes + y
kv_case("c4_1m_rbf_d3_t33", "rbf", 1_000_000, 3, 33, 0.25)

# ---- C5 share: multitask Kronecker MVM, n = 2e5, d = 6, T = 4, 16 probes + y
n5, T = 200_000, 4
X5, _ = synth(n5, 6)
xp5 = B.prep_points("rbf", X5, torch.tensor(0.5), X5.mean(0))
Bf = torch.randn(T, 1, generator=torch.Generator().manual_seed(2)).to(dev)
ktt = Bf @ Bf.t() + 0.5 * torch.eye(T, device=dev)
V5 = torch.randn(17, B.round_up(n5 * T, 4), device=dev)
ms, _ = timed(lambda: kron_matvec(xp5, xp5, ktt, V5))
out["c5_200k_T4_t17"] = dict(n=n5, T=T, cols=17, kron_mvm_ms=ms, tflops=2.0 * n5 * n5 * 17 * T / ms / 1e9, gram=bool(B.kv_flags(xp5, xp5, 68)))
print("c5", out["c5_200k_T4_t17"], flush=True)

doc = json.dumps(out, indent=1)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(doc)
print(doc)
_ = S
