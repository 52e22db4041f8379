"""One- and two-column fused products per family and shape (the mean-cache CG / single-vector Lanczos products of the prediction caches): time per
product and deviation from float64 rows.  GPAMD_LIBRARY selects an A/B build.  python scripts/kv_few_cols_timing.py <tag> -> gpurun_out/kv_few_cols_timing_<tag>.json"""
import json
import os
import sys

import torch

sys.path.insert(0, ".")
from gpytorch_amd import backend as B  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "x"
dev = torch.device("cuda:0")
out = []
for kind, n, d, ls, par in (("matern52", 217_437, 3, 0.35, None), ("matern52", 500_000, 10, 0.8, None), ("matern32", 217_437, 3, 0.35, None), ("matern32", 300_000, 6, 0.5, None),
                            ("rq", 217_437, 3, 0.35, 1.5), ("rbf", 500_000, 3, 0.25, None), ("rbf", 217_437, 3, 0.35, None)):
    g = torch.Generator().manual_seed(0)
    X = torch.rand(n, d, generator=g).to(dev)
    xp = B.prep_points(kind, X, torch.tensor([ls]), X.mean(0), param=par)
    rows = torch.randint(0, n, (48,), generator=g).to(dev)
    Kr = B.kernel_rows(xp, rows, xp).double()      # (the library's own float32 row entries: a consistency reference; the float64 parity lives in tests/test_gpu_kv.py)
    for t in (1, 2, 3):
        vt = torch.randn(t, B.round_up(n, 4), device=dev)
        vt[:, n:] = 0
        res = B.kv(xp, xp, vt)
        ref = (Kr @ vt[:, :n].double().t())
        err = float((res[:, rows].t().double() - ref).abs().max() / ref.abs().max())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            B.kv(xp, xp, vt)
        e1.record()
        torch.cuda.synchronize()
        rec = {"kind": kind, "n": n, "d": d, "t": t, "ms": e0.elapsed_time(e1) / reps, "rel_dev_vs_own_rows": err}
        print(rec, flush=True)
        out.append(rec)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/kv_few_cols_timing_{tag}.json", "w"), indent=1)
