"""Which formulation of the float64 preconditioner apply is fast for which shape (t right-hand sides, rank k, n = 500 000)?"""
import sys
import time

import torch

dev = torch.device("cuda:0")
n = 500_000


def clock(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for t in (1, 2, 11, 65):
    for k in (15, 100):
        r = torch.randn(t, n, device=dev)
        q = torch.randn(k, n, device=dev, dtype=torch.float64)
        r64 = r.double()
        res = {}
        res["to_f64"] = clock(lambda: r.double())
        res["A r@qT"] = clock(lambda: r64 @ q.t())
        res["B (q@rT)T"] = clock(lambda: (q @ r64.t()).t())
        w = r64 @ q.t()
        res["C addmm"] = clock(lambda: torch.addmm(r64, w, q, alpha=-1.0))
        res["D r-(w@q)"] = clock(lambda: r64 - w @ q)
        res["E (qT@wT)T"] = clock(lambda: r64 - (q.t() @ w.t()).t())
        res["F einsum"] = clock(lambda: torch.einsum("tn,kn->tk", r64, q))
        print(t, k, {a: round(b, 3) for a, b in res.items()}, flush=True)
