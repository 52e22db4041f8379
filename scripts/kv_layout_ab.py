"""Times the fused K*V across the column ladder at n = 500 000 (HIP events, both contraction paths) -- run once per library build
(GPAMD_LIBRARY) for A/B comparisons on ONE box.  Usage: python scripts/kv_layout_ab.py <tag> [n]  -> gpurun_out/kv_layout_ab_<tag>.json"""
import json
import os
import sys

import torch

sys.path.insert(0, ".")
from gpytorch_amd import backend as B  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "x"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
dev = torch.device("cuda:0")


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


out = []
for kind, d, ls, ts in (("rbf", 3, 0.25, (1, 2, 4, 11, 17, 33, 65)), ("matern52", 10, 0.8, (1, 11, 65))):
    X = torch.rand(n, d, generator=torch.Generator().manual_seed(0)).to(dev)
    xp = B.prep_points(kind, X, torch.tensor(ls), X.mean(0))
    for t in ts:
        V = torch.randn(t, B.round_up(n, 4), device=dev)
        rec = dict(kind=kind, d=d, t=t, n=n, library=os.environ.get("GPAMD_LIBRARY", "product"))
        for split in (True, False):
            B.SPLIT_CONTRACTION = split
            rec["split_ms" if split else "f32_ms"] = timed(lambda: B.kv(xp, xp, V))
        B.SPLIT_CONTRACTION = None
        print(rec, flush=True)
        out.append(rec)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/kv_layout_ab_{tag}.json", "w"), indent=1)
