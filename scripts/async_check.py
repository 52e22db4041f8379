"""A/B of the DMA-staged double-buffered Gram kernel (flags GRAM|ASYNC) against the synchronous one (GRAM):
equality of results on ragged shapes, then timing.  Usage: python scripts/async_check.py"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from gpytorch_amd import backend as B  # noqa: E402

dev = torch.device("cuda:0")
out = {"equal": [], "timing": []}
for kind, n, m, d, t in [("rbf", 700, 1100, 3, 33), ("rbf", 1025, 1300, 2, 64), ("rbf", 600, 2100, 3, 65), ("matern52", 999, 3001, 3, 65),
                         ("rbf", 257, 128, 1, 64), ("rbf", 5000, 4097, 4, 40), ("matern32", 3000, 2999, 3, 50), ("rbf", 300, 31, 3, 65), ("rbf", 300, 129, 2, 33)]:
    g = torch.Generator().manual_seed(n + m)
    X1 = torch.rand(n, d, generator=g).to(dev)
    X2 = torch.rand(m, d, generator=g).to(dev)
    ls = torch.tensor(0.4)
    sh = X1.mean(0)
    p1, p2 = B.prep_points(kind, X1, ls, sh), B.prep_points(kind, X2, ls, sh)
    vt = torch.randn(t, B.round_up(m, 4), device=dev)
    vt[:, m:] = 0
    B.FORCE_KV_FLAGS = 1
    a = B.kv(p1, p2, vt).clone()
    B.FORCE_KV_FLAGS = 3
    b = B.kv(p1, p2, vt).clone()
    B.FORCE_KV_FLAGS = None
    err = float((a[:, :n] - b[:, :n]).abs().max() / a[:, :n].abs().max())
    nbad = int(((a[:, :n] - b[:, :n]).abs() > 1e-6 * a[:, :n].abs().max()).sum())
    print(kind, n, m, d, t, "max rel diff", err, "elements off:", nbad, "of", a[:, :n].numel(), flush=True)
    out["equal"].append([kind, n, m, d, t, err])
for n, t in [(100_000, 65), (500_000, 65), (200_000, 64)]:
    X = torch.rand(n, 3, device=dev)
    xp = B.prep_points("rbf", X, torch.tensor(0.25), X.mean(0))
    vt = torch.randn(t, B.round_up(n, 4), device=dev)
    for flags in (1, 3):
        B.FORCE_KV_FLAGS = flags
        B.kv(xp, xp, vt)
        torch.cuda.synchronize()
        reps = 3 if n >= 400_000 else 8
        t0 = time.perf_counter()
        for _ in range(reps):
            B.kv(xp, xp, vt)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        S, jc, _ = B.kv_plan("rbf", n, n, 3, t, flags, B.round_up(n, 4))
        rec = dict(n=n, t=t, flags=flags, ms=ms, tflops=2.0 * n * n * t / ms / 1e9, S=S, jchunk=jc)
        print(rec, flush=True)
        out["timing"].append(rec)
    B.FORCE_KV_FLAGS = None
import os
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/async_check.json", "w"), indent=1)
