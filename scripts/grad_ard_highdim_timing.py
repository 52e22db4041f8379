"""ARD backward at d >= 8: the split W contraction in the per-dimension mode (spills: 44-182 VGPRs at two waves per SIMD) against the fp32 W
contraction the library keeps there (backend.GRAD_SPLIT_MAX_ARD_DIM).  Usage: python scripts/grad_ard_highdim_timing.py [n] -> gpurun_out/grad_ard_highdim.json"""
import json
import os
import sys

import torch

sys.path.insert(0, ".")
from gpytorch_amd import backend as B  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
dev = torch.device("cuda:0")
t = 65


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = []
for kind, d, ls in (("matern52", 10, 0.8), ("rbf", 8, 0.7), ("rbf", 16, 1.2)):
    X = torch.rand(n, d, generator=torch.Generator().manual_seed(0)).to(dev)
    xp = B.prep_points(kind, X, torch.full((d,), ls), X.mean(0))
    lt = torch.randn(t, B.round_up(n, 4), device=dev).abs_()
    rt = torch.randn(t, B.round_up(n, 4), device=dev).abs_()
    rec = dict(kind=kind, d=d, n=n, t=t)
    res = {}
    for name, maxd in (("fp32W", 6), ("splitW", 16)):
        B.GRAD_SPLIT_MAX_ARD_DIM = maxd
        rec[name + "_ard_ms"] = timed(lambda: B.kv_grad2(xp, xp, lt, rt, iso=False))
        res[name] = B.kv_grad2(xp, xp, lt, rt, iso=False)[0].double().cpu()
    B.GRAD_SPLIT_MAX_ARD_DIM = 6
    rec["rel_dev_split_vs_fp32"] = float(((res["splitW"] - res["fp32W"]).abs() / res["fp32W"].abs().clamp_min(1e-30))[: 1 + d].max())
    rec["kv_ms"] = timed(lambda: B.kv(xp, xp, rt))
    print(rec, flush=True)
    out.append(rec)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/grad_ard_highdim.json", "w"), indent=1)
