"""mBCG wall time per solve with and without the hipGraph replay (settings.cg_graph) on launch-bound problems:
python scripts/cg_graph_timing.py [tag] -> gpurun_out/cg_graph_<tag>.json"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from gpytorch_amd import backend as B  # noqa: E402
from gpytorch_amd import settings  # noqa: E402
from gpytorch_amd.bbmm import build_preconditioner  # noqa: E402
from gpytorch_amd.linear_cg import linear_cg  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "x"
dev = torch.device("cuda:0")
out = []
for n, t, rank in [(2000, 11, 0), (2000, 11, 15), (5000, 11, 0), (10000, 11, 15), (20000, 11, 0), (5000, 65, 0)]:
    g = torch.Generator().manual_seed(n)
    X = torch.rand(n, 3, generator=g)
    xp = B.prep_points("rbf", X.to(dev), torch.tensor(0.25), X.mean(0).to(dev))
    rt = torch.zeros(t, B.round_up(n, 4), device=dev)
    rt[:, :n] = torch.randn(t, n, generator=g).to(dev)
    scale, noise = torch.tensor([1.0], device=dev), torch.tensor([0.01], device=dev)
    pc = build_preconditioner(xp, scale, noise, rank=rank, min_size=0) if rank else None
    rec = dict(n=n, t=t, precond_rank=rank)
    for graph in (False, True):
        with settings.cg_graph(graph):
            linear_cg(xp, scale, noise, rt, n_tridiag=t - 1, tolerance=1e-2, max_iter=1000, preconditioner=pc)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                _, info = linear_cg(xp, scale, noise, rt, n_tridiag=t - 1, tolerance=1e-2, max_iter=1000, preconditioner=pc)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
        rec["graph_ms" if graph else "eager_ms"] = ms
        rec["iterations"] = info.iterations
    rec["speedup"] = rec["eager_ms"] / rec["graph_ms"]
    rec["eager_us_per_iteration"] = 1e3 * rec["eager_ms"] / rec["iterations"]
    rec["graph_us_per_iteration"] = 1e3 * rec["graph_ms"] / rec["iterations"]
    print(json.dumps(rec), flush=True)
    out.append(rec)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/cg_graph_{tag}.json", "w"), indent=1)
