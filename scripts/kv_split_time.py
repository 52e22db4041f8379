"""Times the fused K*V with and without the split-operand contraction (GPAMD_KV_SPLIT) at the bench shapes and checks a row
sample of each against the float64 oracle: python scripts/kv_split_time.py [tag] -> gpurun_out/kv_split_<tag>.json"""
import json
import os
import sys

import torch

sys.path.insert(0, ".")
from gpytorch_amd import backend as B  # noqa: E402
from oracle import kernels as OK  # noqa: E402  (checker only)

tag = sys.argv[1] if len(sys.argv) > 1 else "x"
dev = torch.device("cuda:0")
out = []


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


cases = [("rbf", 500_000, 3, 0.25, [65, 64, 33, 32, 17, 11, 5]), ("matern52", 500_000, 10, 0.8, [65]), ("rbf", 100_000, 3, 0.25, [65])]
if len(sys.argv) > 2:
    cases = [("rbf", int(sys.argv[2]), 3, 0.25, [int(v) for v in sys.argv[3].split(",")])]
for kind, n, d, ls, ts in cases:
    g = torch.Generator().manual_seed(0)
    X = torch.rand(n, d, generator=g)
    xp = B.prep_points(kind, X.to(dev), torch.tensor(ls), X.mean(0).to(dev))
    for t in ts:
        V = torch.randn(t, B.round_up(n, 4), generator=g)
        vt = V.to(dev)
        rec = dict(kind=kind, n=n, d=d, t=t)
        rows = torch.arange(0, n, max(1, n // 512))[:512]
        Kr = (OK.rbf(X[rows].double(), X.double(), ls, direct=True) if kind == "rbf" else None)
        for split in (0, 1):
            B.SPLIT_CONTRACTION = bool(split)
            ms = timed(lambda: B.kv(xp, xp, vt))
            rec[f"ms_split{split}"] = ms
            rec[f"tflops_split{split}"] = 2.0 * n * n * t / ms / 1e9
            if Kr is not None:
                o = B.kv(xp, xp, vt)[:, rows].double().cpu()
                ref = (Kr @ V[:, :n].double().T).T
                rec[f"relerr_split{split}"] = float(((o - ref).abs().amax(1) / ref.abs().amax(1)).max())
        print(json.dumps(rec), flush=True)
        out.append(rec)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/kv_split_{tag}.json", "w"), indent=1)
