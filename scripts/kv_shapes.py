"""Times the fused K*V at the shapes that matter (bench shape, configs C2/C3/C4 column counts, small t) and checks each
against a float64 row sample.  Usage: python scripts/kv_shapes.py [tag]   -> gpurun_out/kv_shapes_<tag>.json"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from gpytorch_amd import backend as B  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "x"
dev = torch.device("cuda:0")
CASES = [
    ("rbf", 100_000, 3, 65, 0.25), ("rbf", 500_000, 3, 65, 0.25), ("matern52", 200_000, 10, 65, 0.8),
    ("rbf", 200_000, 3, 33, 0.25), ("rbf", 200_000, 3, 129, 0.25), ("matern32", 200_000, 6, 65, 0.5),
    ("rbf", 200_000, 16, 65, 1.2), ("rbf", 500_000, 3, 11, 0.25), ("rbf", 500_000, 3, 16, 0.25), ("rbf", 500_000, 3, 17, 0.25),
    ("matern52", 200_000, 10, 11, 0.8), ("rbf", 500_000, 3, 1, 0.25), ("rbf", 500_000, 3, 4, 0.25), ("matern52", 500_000, 3, 1, 0.25),
]
out = []
for kind, n, d, t, ls in CASES:
    g = torch.Generator(device="cpu").manual_seed(0)
    X = torch.rand(n, d, generator=g).to(dev)
    xp = B.prep_points(kind, X, torch.tensor(ls), X.mean(0))
    vt = torch.randn(t, B.round_up(n, 4), device=dev)
    vt[:, n:] = 0
    res = B.kv(xp, xp, vt)
    torch.cuda.synchronize()
    reps = 3 if n >= 400_000 else 6
    t0 = time.perf_counter()
    for _ in range(reps):
        B.kv(xp, xp, vt)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    # float64 check on 64 sampled rows
    rows = torch.randint(0, n, (64,), generator=g).to(dev)
    x64 = xp.xp.double()
    dist2 = (x64[rows].unsqueeze(1) - x64.unsqueeze(0)).pow(2).sum(-1)
    if kind == "rbf":
        K = torch.exp2(-dist2)
    else:
        r = dist2.sqrt()
        K = {"matern12": torch.exp(-r), "matern32": (1 + r) * torch.exp(-r), "matern52": (1 + r + dist2 / 3) * torch.exp(-r)}[kind]
    ref = K @ vt[:, :n].double().t()
    err = float((res[:, rows].t().double() - ref).abs().max() / ref.abs().max())
    rec = dict(kind=kind, n=n, d=d, t=t, ms=ms, tflops=2.0 * n * n * t / ms / 1e9, gram=bool(B.kv_flags(xp, xp, t)), rel_err=err)
    print(rec, flush=True)
    out.append(rec)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/kv_shapes_{tag}.json", "w"), indent=1)
