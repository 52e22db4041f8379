"""Batch of small exact GPs: one marginal-log-likelihood evaluation + backward, stacked path (gpytorch_amd/batched.py) against the launch
plan over members.  Writes gpurun_out/batch_small_timing.json.
Usage: python scripts/batch_small_timing.py [b[+b2+...]] [n[+n2+...]] [d]     (lists joined by '+': every (b, n) pair is timed)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpytorch_amd as g  # noqa: E402

bs_list = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "64").split("+")]
ns_list = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "200").split("+")]
d = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")


def run(b, n):
    bs = torch.Size([b])
    gen = torch.Generator().manual_seed(0)
    X = torch.rand(b, n, d, generator=gen).to(dev)
    Y = (torch.sin(3 * X.sum(-1).cpu()) + 0.1 * torch.randn(b, n, generator=gen)).to(dev)

    class M(g.models.ExactGP):
        def __init__(self, x, y, lik):
            super().__init__(x, y, lik)
            self.mean_module = g.means.ConstantMean(batch_shape=bs)
            self.covar_module = g.kernels.ScaleKernel(g.kernels.RBFKernel(batch_shape=bs), batch_shape=bs)

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    lik = g.likelihoods.GaussianLikelihood(batch_shape=bs).to(dev)
    m = M(X, Y, lik).to(dev)
    mll = g.ExactMarginalLogLikelihood(lik, m)
    m.train()
    lik.train()
    out = {"b": b, "n": n, "d": d}
    for stacked in (True, False):
        with g.settings.batched_small_members(stacked), g.settings.max_cholesky_size(max(n, 800)):
            ts = []
            for _ in range(4):
                m.zero_grad()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                val = mll(m(X), Y).sum()
                val.backward()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            out["stacked" if stacked else "member_loop"] = {"ms": [round(1e3 * t, 2) for t in ts], "mll_sum": float(val.detach())}
    out["speedup_last"] = round(out["member_loop"]["ms"][-1] / out["stacked"]["ms"][-1], 2)
    return out


results = [run(b, n) for b in bs_list for n in ns_list]
os.makedirs("gpurun_out", exist_ok=True)
json.dump(results if len(results) > 1 else results[0], open("gpurun_out/batch_small_timing.json", "w"), indent=1)
for r in results:
    print(json.dumps(r))
