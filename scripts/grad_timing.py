"""Times the fused bilinear derivative (backward of one MLL evaluation) at the bench shape against one K*V of equal flops:
Gram-form kernel (kv_grad2, single lengthscale / ARD / ARD + input gradients) vs the direct-difference kernel (kv_grad).
Usage: python scripts/grad_timing.py [tag] [n] [t]  -> gpurun_out/grad_timing_<tag>.json"""
import json
import os
import sys

import torch

sys.path.insert(0, ".")
from gpytorch_amd import backend as B  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "x"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
t = int(sys.argv[3]) if len(sys.argv) > 3 else 65
dev = torch.device("cuda:0")


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
for kind, d, ls in (("rbf", 3, 0.25), ("matern52", 10, 0.8)):
    X = torch.rand(n, d, generator=torch.Generator().manual_seed(0)).to(dev)
    xp = B.prep_points(kind, X, torch.tensor(ls), X.mean(0))
    lt = torch.randn(t, B.round_up(n, 4), device=dev)
    rt = torch.randn(t, B.round_up(n, 4), device=dev)
    flop = 2.0 * n * n * t
    rec = dict(kind=kind, n=n, d=d, t=t)
    rec["kv_ms"] = timed(lambda: B.kv(xp, xp, rt))
    rec["grad_direct_iso_ms"] = timed(lambda: B.kv_grad(xp, xp, lt, rt, iso=True))
    rec["grad_direct_ard_ms"] = timed(lambda: B.kv_grad(xp, xp, lt, rt, iso=False))
    for split in (False, True):   # W = L^T R on the fp32 MFMAs / on hi-lo split f16 operands (kv_grad2.hpp WSPLIT)
        B.SPLIT_CONTRACTION = split
        sfx = "_split" if split else ""
        rec["grad2_iso%s_ms" % sfx] = timed(lambda: B.kv_grad2(xp, xp, lt, rt, iso=True))
        rec["grad2_ard%s_ms" % sfx] = timed(lambda: B.kv_grad2(xp, xp, lt, rt, iso=False))
        rec["grad2_ard_xgrad%s_ms" % sfx] = timed(lambda: B.kv_grad2(xp, xp, lt, rt, iso=False, want_gz1=True))
        # accuracy on NON-cancelling vectors (|randn|): random-sign vectors make the n^2-term sums cancel by ~1e4, which turns the 2^-22
        # operand precision of either contraction into an apparent 1e-3 "deviation" between any two summation orders
        la, ra = lt.abs(), rt.abs()
        a = B.kv_grad(xp, xp, la, ra, iso=False)
        b, _ = B.kv_grad2(xp, xp, la, ra, iso=False)
        rec["max_rel_dev_vs_direct" + sfx] = float(((a - b[: a.numel()]).abs() / a.abs().clamp_min(1e-30))[: 1 + d].max())
        b0, _ = B.kv_grad2(xp, xp, la, ra, iso=True)
        rec["iso_rel_dev_vs_direct" + sfx] = float(abs(float(a[1 : 1 + d].sum()) - float(b0[1])) / abs(float(a[1 : 1 + d].sum())))
    B.SPLIT_CONTRACTION = None
    for k in list(rec):
        if k.endswith("_ms"):
            rec[k.replace("_ms", "_tflops")] = flop / rec[k] / 1e9
    print(rec, flush=True)
    out.append(rec)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/grad_timing_{tag}.json", "w"), indent=1)
