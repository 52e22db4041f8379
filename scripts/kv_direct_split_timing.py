"""Round 5: direct-difference products with the contraction on the VALU / fp32 MFMAs (flags 0: kv_valu, kv_mfma) against the split contraction
(flags GPAMD_KV_SPLIT alone: kv_directh.hpp).  ms per product by HIP events, relative error of 256 sampled rows against float64.
Usage: python scripts/kv_direct_split_timing.py [out.json]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpytorch_amd import backend as B  # noqa: E402

dev = torch.device("cuda:0")
out = []
for kind, n, d, t, ls in [("matern52", 217_437, 3, 11, 0.05), ("rbf", 217_437, 3, 11, 0.05), ("matern52", 217_437, 3, 32, 0.05), ("matern52", 217_437, 3, 33, 0.05), ("matern52", 217_437, 3, 64, 0.05), ("matern52", 217_437, 3, 65, 0.05), ("matern52", 100_000, 8, 65, 0.2),
                          ("rbf", 100_000, 10, 11, 0.3), ("matern52", 100_000, 6, 11, 0.2), ("matern12", 217_437, 3, 11, 0.3), ("rbf", 500_000, 3, 11, 0.02)]:
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.rand(n, d, device=dev, generator=g)
    xp = B.prep_points(kind, X, torch.tensor(ls), X.mean(0))
    vt = torch.randn(t, B.round_up(n, 4), device=dev, generator=g)
    rows = torch.randint(0, n, (256,), device=dev, generator=g)
    z = xp.xp[:, :d].double()
    S = (z[rows].unsqueeze(1) - z.unsqueeze(0)).pow(2).sum(-1)
    if kind == "rbf":
        K = torch.exp2(-S)
    elif kind == "matern12":
        K = torch.exp(-S.sqrt())
    else:
        r = S.sqrt()
        K = (1.0 + r + S / 3.0) * torch.exp(-r)
    ref = K @ vt[:, :n].double().t()
    rec = dict(kind=kind, n=n, d=d, t=t, lengthscale=ls)
    for name, flags in (("contraction_on_valu_or_fp32_mfma", 0), ("split_contraction", B.KV_SPLIT)):
        B.FORCE_KV_FLAGS = flags
        try:
            res = B.kv(xp, xp, vt)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                B.kv(xp, xp, vt)
            e1.record()
            torch.cuda.synchronize()
        finally:
            B.FORCE_KV_FLAGS = None
        rec[name + "_ms"] = e0.elapsed_time(e1) / 5
        rec[name + "_rel_err"] = float((res[:, rows].t().double() - ref).abs().max() / ref.abs().max())
    rec["speedup"] = rec["contraction_on_valu_or_fp32_mfma_ms"] / rec["split_contraction_ms"]
    rec["pairs_per_second_split"] = n * n / rec["split_contraction_ms"] * 1e3
    print(rec, flush=True)
    out.append(rec)
path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/kv_direct_split_timing.json"
os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
json.dump(out, open(path, "w"), indent=1)
