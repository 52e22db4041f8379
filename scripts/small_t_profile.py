"""Times one K*V at small column counts (the t = 1 products that bound the cold posterior: mean-cache CG and the
LOVE Lanczos recurrence) and the two posterior caches themselves.  Usage: python scripts/small_t_profile.py [n] [d]"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from gpytorch_amd import backend as B  # noqa: E402
from gpytorch_amd import settings as S  # noqa: E402
from gpytorch_amd.lanczos import lanczos_tridiag  # noqa: E402
from gpytorch_amd.linear_cg import linear_cg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
torch.manual_seed(0)
X = torch.rand(n, d, device=dev)
y = torch.sin(6.2831853 * X[:, 0]) + torch.cos(3.14159265 * X.sum(-1)) + 0.1 * torch.randn(n, device=dev)
xp = B.prep_points("rbf", X, torch.tensor(0.25), X.mean(0))
sc = torch.ones(1, device=dev)
nz = torch.full((1,), 0.1, device=dev)
out = {"n": n, "d": d}
for t in (1, 2, 4, 8, 9, 16, 32):
    vt = torch.randn(t, B.round_up(n, 4), device=dev)
    B.kv(xp, xp, vt, scale=sc, dscale=nz, vd=vt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        B.kv(xp, xp, vt, scale=sc, dscale=nz, vd=vt)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    out[f"kv_t{t}_ms"] = ms
    out[f"kv_t{t}_Tpairs_per_s"] = n * n / ms / 1e9
    print(t, ms, flush=True)
rhs = B.to_probe_major(y.unsqueeze(-1))
torch.cuda.synchronize()
t0 = time.perf_counter()
sol, info = linear_cg(xp, sc, nz, rhs, tolerance=S.eval_cg_tolerance.value(), max_iter=1000)
torch.cuda.synchronize()
out["mean_cg_s"] = time.perf_counter() - t0
out["mean_cg_iters"] = info.iterations
print(out["mean_cg_s"], info.iterations, flush=True)
t0 = time.perf_counter()
Q, T = lanczos_tridiag(xp, sc, nz, 100)
torch.cuda.synchronize()
out["lanczos100_s"] = time.perf_counter() - t0
out["lanczos_rank"] = int(T.shape[0])
print(json.dumps(out))
import os
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/small_t_profile.json", "w"), indent=1)
