"""The split count S of the compact launch of a protein-shaped product (34 304 x 36 584 pairs, 11 columns, d = 9): time per launch against S,
beside the planner's own choice (csrc/api.hip plan_split).  -> gpurun_out/kv_small_split_sweep.json"""
import json
import os
import sys

import torch

sys.path.insert(0, ".")
from gpytorch_amd import backend as B  # noqa: E402

dev = torch.device("cuda:0")
out = []
for (n, m, d, t, kind) in ((34_304, 36_584, 9, 11, "rbf"), (34_304, 36_584, 9, 1, "rbf"), (65_536, 65_536, 3, 11, "rbf"), (100_000, 100_000, 10, 11, "matern52")):
    g = torch.Generator().manual_seed(0)
    X = (torch.randn(m, d, generator=g) * 0.5).clamp_(-1.5, 1.5).to(dev)
    ls = torch.tensor([1.0])
    x2 = B.prep_points(kind, X, ls, X.mean(0))
    x1 = B.prep_points(kind, X[:n].contiguous(), ls, X.mean(0))
    assert B.gram_mode(x1, x2) == 1, B.gram_mode(x1, x2)
    vt = torch.randn(t, B.round_up(m, 4), device=dev)
    vt[:, m:] = 0
    flags = B.kv_flags(x1, x2, t)
    ldo = B.round_up(n, 4)
    plan = B.kv_plan.__wrapped__ if hasattr(B.kv_plan, "__wrapped__") else B.kv_plan
    S0, jc0, ws0 = plan(kind, n, m, d, t, flags, ldo)
    ref = B.kv(x1, x2, vt).clone()
    extra = ws0 - S0 * t * ldo

    def clock(reps=30):
        B.kv(x1, x2, vt)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            B.kv(x1, x2, vt)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    rec = {"shape": [n, m, d, t, kind], "planner": {"S": S0, "jchunk": jc0, "us_per_product": clock()}, "sweep": []}
    orig = B.kv_plan
    for S in range(1, 41):
        jc = ((m + S - 1) // S + 127) // 128 * 128
        if (m + jc - 1) // jc != S:
            continue
        B.kv_plan = lambda *a, S=S, jc=jc: (S, jc, S * t * ldo + extra + 4096)
        us = clock()
        err = float((B.kv(x1, x2, vt) - ref).abs().max() / ref.abs().max())
        rec["sweep"].append({"S": S, "jchunk": jc, "units_of_512_rows": -(-n // 512) * S, "us_per_product": us, "rel_dev_from_planner_result": err})
    B.kv_plan = orig
    best = min(rec["sweep"], key=lambda r: r["us_per_product"])
    rec["best"] = best
    print(rec["shape"], "planner", rec["planner"], "best", best, flush=True)
    print("   ", [(r["S"], round(r["us_per_product"])) for r in rec["sweep"]], flush=True)
    out.append(rec)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/kv_small_split_sweep.json", "w"), indent=1)
