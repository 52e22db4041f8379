"""Round 5: the fused float64 K*V after the lean exp2 / exp / sqrt (csrc/common.hpp) and the VALU contraction for <= 4 columns (kv_f64v_kernel).
Per shape: ms per product (HIP events, 5 launches), TFLOP/s, fraction of the measured float64 MFMA rate (72.07 TF), and the largest relative error of
256 sampled output rows against dense float64 torch.  Usage: python scripts/f64_gen_timing.py [out.json]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpytorch_amd import backend as B  # noqa: E402

dev = torch.device("cuda:0")
PEAK = 72.07
out = []
for kind, n, d, t in [("rbf", 100_000, 3, 65), ("rbf", 100_000, 3, 1), ("rbf", 100_000, 3, 2), ("rbf", 100_000, 3, 4), ("rbf", 100_000, 3, 11),
                      ("matern52", 100_000, 3, 65), ("matern52", 100_000, 3, 1), ("rbf", 50_000, 10, 65), ("matern52", 50_000, 10, 65), ("rbf", 50_000, 16, 65)]:
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.rand(n, d, device=dev, dtype=torch.float64, generator=g)
    ls = 0.25 if d <= 3 else 0.8
    xp = B.prep_points(kind, X, torch.tensor(ls, dtype=torch.float64), X.mean(0))
    assert B.fused_f64(xp, xp)
    vt = torch.randn(t, B.round_up(n, 4), device=dev, dtype=torch.float64, generator=g)
    res = B.kv(xp, xp, vt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        B.kv(xp, xp, vt)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    rows = torch.randint(0, n, (256,), device=dev, generator=g)
    z = xp.xp[:, :d]
    S = (z[rows].unsqueeze(1) - z.unsqueeze(0)).pow(2).sum(-1)
    if kind == "rbf":
        K = torch.exp2(-S)
    else:
        r = S.sqrt()
        K = (1.0 + r + S / 3.0) * torch.exp(-r)
    ref = K @ vt[:, :n].t()
    got = res[:, rows].t()
    err = float((got - ref).abs().max() / ref.abs().max())
    rec = dict(kind=kind, n=n, d=d, t=t, ms=ms, tflops_f64=2.0 * n * n * t / ms / 1e9, frac_of_measured_mfma_f64_rate=2.0 * n * n * t / ms / 1e9 / PEAK,
               pairs_per_second=n * n / ms * 1e3, max_rel_err_vs_dense_float64=err)
    print(rec, flush=True)
    out.append(rec)
path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/f64_gen_timing.json"
os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
json.dump(out, open(path, "w"), indent=1)
