"""Where the cold predictive posterior spends its time at the metric size (n = 500 000, d = 3, RBF, 10 000 test points):
mean-cache solve (mBCG at eval_cg_tolerance 0.01) without / with the pivoted-Cholesky preconditioner (reference default rank
15, and rank 100), the 100-step Lanczos root-inverse (LOVE cache), and the K_*X products.
Usage: python scripts/posterior_profile.py [tag] [n]  -> gpurun_out/posterior_profile_<tag>.json"""
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from gpytorch_amd import backend as B  # noqa: E402
from gpytorch_amd import settings as S  # noqa: E402
from gpytorch_amd.bbmm import build_preconditioner  # noqa: E402
from gpytorch_amd.lanczos import root_inv_decomposition  # noqa: E402
from gpytorch_amd.linear_cg import linear_cg  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "x"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
X = torch.rand(n, 3, generator=g)
y = torch.sin(2 * math.pi * X[:, 0]) + torch.cos(math.pi * X.sum(-1)) + 0.1 * torch.randn(n, generator=g)
Xd, yd = X.to(dev), y.to(dev)
xp = B.prep_points("rbf", Xd, torch.tensor([0.25]), Xd.mean(0))
sc, s2 = torch.tensor([1.0], device=dev), torch.tensor([0.1], device=dev)
rhs = B.to_probe_major(yd.unsqueeze(-1))


def clock(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) * 1e3


out = {"n": n}
B.kv(xp, xp, rhs)  # warm
sols = {}
for rank in (0, 15, 100):
    pre, ms_pre = clock(lambda: build_preconditioner(xp, sc, s2, rank=rank, min_size=0) if rank else None)
    (sol, info), ms = clock(lambda: linear_cg(xp, sc, s2, rhs, tolerance=0.01, max_iter=2000, preconditioner=pre))
    res = B.kv(xp, xp, sol, scale=sc, dscale=s2, vd=sol)[:, :n] - rhs[:, :n]
    out[f"mean_cache_precond{rank}"] = dict(build_ms=ms_pre, solve_ms=ms, iterations=info.iterations, reached=info.tolerance_reached,
                                           true_rel_residual=float(res.norm() / rhs[:, :n].norm()))
    sols[rank] = sol
    print(rank, out[f"mean_cache_precond{rank}"], flush=True)
out["mean_cache_agreement_rel"] = float((sols[100] - sols[0])[:, :n].norm() / sols[0][:, :n].norm())
_, ms = clock(lambda: root_inv_decomposition(xp, sc, s2, max_iter=100, generator=torch.Generator(device=dev).manual_seed(1)))
out["lanczos_100_steps_ms"] = ms
Xs = torch.rand(10_000, 3, generator=g).to(dev)
xs = B.prep_points("rbf", Xs, torch.tensor([0.25]), Xd.mean(0))
_, out["test_mean_product_ms"] = clock(lambda: B.kv(xs, xp, sols[0]))
rt = torch.randn(100, B.round_up(n, 4), device=dev)
_, out["test_love_product_100cols_ms"] = clock(lambda: B.kv(xs, xp, rt))
print(out)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/posterior_profile_{tag}.json", "w"), indent=1)
