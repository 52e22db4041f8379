"""A/B of the 4-column-group contraction kernel (kv_gram4, default for 3..32 columns) against the previous selection
(VALU contraction for t <= 16, 32-column MFMA tile above; flag KV_WIDE) at n = 500 000, each checked against a float64 row
sample.  Usage: python scripts/kv_small_t.py [tag] [n]   -> gpurun_out/kv_small_t_<tag>.json"""
import json
import os
import sys

import torch

sys.path.insert(0, ".")
from gpytorch_amd import backend as B  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "x"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
dev = torch.device("cuda:0")
PEAK = 157.3
CASES = [("rbf", 3, t, 0.25) for t in (1, 2, 3, 4, 8, 11, 12, 16, 17, 20, 24, 32, 33)] + [("matern52", 10, 11, 0.8), ("matern52", 10, 16, 0.8),
                                                                                           ("matern32", 6, 11, 0.5), ("rbf", 16, 11, 1.2)]
out = []
for kind, d, t, ls in CASES:
    g = torch.Generator(device="cpu").manual_seed(0)
    X = torch.rand(n, d, generator=g).to(dev)
    xp = B.prep_points(kind, X, torch.tensor(ls), X.mean(0))
    vt = torch.randn(t, B.round_up(n, 4), device=dev)
    vt[:, n:] = 0
    rows = torch.randint(0, n, (64,), generator=g).to(dev)
    x64 = xp.xp.double()
    dist2 = (x64[rows].unsqueeze(1) - x64.unsqueeze(0)).pow(2).sum(-1)
    if kind == "rbf":
        K = torch.exp2(-dist2)
    else:
        r = dist2.sqrt()
        K = {"matern32": (1 + r) * torch.exp(-r), "matern52": (1 + r + dist2 / 3) * torch.exp(-r)}[kind]
    ref = K @ vt[:, :n].double().t()
    rec = dict(kind=kind, n=n, d=d, t=t)
    for name, flags in (("new", B.KV_GRAM), ("g4", B.KV_GRAM | B.KV_G4), ("wide", B.KV_GRAM | B.KV_WIDE)):
        B.FORCE_KV_FLAGS = flags
        res = B.kv(xp, xp, vt)
        torch.cuda.synchronize()
        reps = 3
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            B.kv(xp, xp, vt)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        rec[name + "_ms"] = ms
        rec[name + "_tflops"] = 2.0 * n * n * t / ms / 1e9
        rec[name + "_frac_fp32_mfma_peak"] = rec[name + "_tflops"] / PEAK
        rec[name + "_rel_err"] = float((res[:, rows].t().double() - ref).abs().max() / ref.abs().max())
        rec[name + "_pairs_per_s"] = float(n) * n / (ms * 1e-3)
    B.FORCE_KV_FLAGS = None
    print(rec, flush=True)
    out.append(rec)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/kv_small_t_{tag}.json", "w"), indent=1)
