"""Fused float64 K*V (kv_f64.hpp) against the row-block x DGEMM path.  Usage: python scripts/f64_timing.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from gpytorch_amd import backend as B  # noqa: E402

dev = torch.device("cuda:0")
out = []
for n, d, t in [(50_000, 3, 11), (100_000, 3, 65), (100_000, 3, 1), (50_000, 8, 65), (50_000, 10, 65), (50_000, 16, 65), (50_000, 10, 1), (50_000, 16, 11)]:
    X = torch.rand(n, d, device=dev, dtype=torch.float64)
    xp = B.prep_points("rbf", X, torch.tensor(0.25 if d <= 3 else 0.8, dtype=torch.float64), X.mean(0))
    assert B.fused_f64(xp, xp)
    vt = torch.randn(t, B.round_up(n, 4), device=dev, dtype=torch.float64)
    rec = dict(n=n, d=d, t=t)
    for name, force in (("fused_ms", False), ("rowblock_ms", True)):
        if force and n > 50_000:
            continue   # (the row-block path at n = 1e5 takes seconds per product: measured in round 2)
        B.FORCE_CHUNKED = force
        B.kv(xp, xp, vt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            B.kv(xp, xp, vt)
        torch.cuda.synchronize()
        rec[name] = (time.perf_counter() - t0) / 3 * 1e3
    B.FORCE_CHUNKED = False
    rec["fused_tflops_f64"] = 2.0 * n * n * t / rec["fused_ms"] / 1e9
    print(rec, flush=True)
    out.append(rec)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/f64_timing.json", "w"), indent=1)
