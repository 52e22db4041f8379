"""Time of one preconditioner apply (projection + partial sum + fused subtraction, float64) per (n, rank, columns); run once per library
(GPAMD_LIBRARY selects an A/B build of the same ABI).  python scripts/precond_apply_timing.py <tag> -> gpurun_out/precond_apply_timing_<tag>.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpytorch_amd.linear_cg import Preconditioner  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "x"
dev = torch.device("cuda:0")
out = []
for n in (36_584, 100_000, 500_000):
    for k in (15, 100, 256):
        q = torch.linalg.qr(torch.randn(n, k, device=dev, dtype=torch.float64)).Q.t().contiguous()
        pre = Preconditioner(q, torch.tensor([0.1], device=dev), torch.zeros((), device=dev), None)
        for t in (1, 11, 65):
            r = torch.randn(t, n, device=dev)
            o = torch.empty_like(r)
            pre.apply_(r, o)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 50
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                pre.apply_(r, o)
            e1.record()
            torch.cuda.synchronize()
            ref = (r.double() - (r.double() @ q.t()) @ q) / 0.1
            rec = {"n": n, "rank": k, "columns": t, "us_per_apply": e0.elapsed_time(e1) / reps * 1e3, "max_abs_dev_vs_float64": float((o.double() - ref).abs().max())}
            print(rec, flush=True)
            out.append(rec)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/precond_apply_timing_{tag}.json", "w"), indent=1)
