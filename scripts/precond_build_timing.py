"""Pivoted-Cholesky preconditioner build (gpamd_pivoted_cholesky_f32: one launch per pivot step since round 6 + the float64 Cholesky-QR) per (n, rank).
    python scripts/precond_build_timing.py -> gpurun_out/precond_build_timing.json"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from gpytorch_amd import backend as B  # noqa: E402
from gpytorch_amd.bbmm import build_preconditioner  # noqa: E402

dev = torch.device("cuda:0")
rows = []
for n, d, kind, ls, ranks in ((500_000, 3, "rbf", 0.25, (15, 100, 128, 256, 384)), (36_584, 9, "rbf", 1.2, (15, 100)), (217_437, 3, "matern52", 0.2, (15, 100)), (2_000, 3, "rbf", 0.25, (15,))):
    g = torch.Generator().manual_seed(0)
    X = torch.rand(n, d, generator=g).to(dev)
    xp = B.prep_points(kind, X, torch.tensor([ls], device=dev), X.mean(0))
    sc, nz = torch.tensor([1.0], device=dev), torch.tensor([0.1], device=dev)
    for rank in ranks:
        build_preconditioner(xp, sc, nz, rank=rank, min_size=0)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            pre = build_preconditioner(xp, sc, nz, rank=rank, min_size=0)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        # the factor alone
        B.pivoted_cholesky(xp, sc, rank, 1e-3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            lt, _, k = B.pivoted_cholesky(xp, sc, rank, 1e-3)
        torch.cuda.synchronize()
        rows.append({"n": n, "d": d, "kind": kind, "rank": rank, "columns_produced": int(k), "build_ms_min": min(ts), "build_ms_median": sorted(ts)[2],
                     "pivoted_cholesky_ms": (time.perf_counter() - t0) * 1e3 / 5})
        print(rows[-1], flush=True)
json.dump({"rows": rows}, open("gpurun_out/precond_build_timing.json", "w"), indent=1)
