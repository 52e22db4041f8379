"""Batches of MID-SIZE exact GPs (800 < n <= 4000 per member): one MLL evaluation + backward on
  (a) the launch plan over members, each on the BBMM path (the library default above max_cholesky_size = 800), and
  (b) the stacked dense path of gpytorch_amd/batched.py with max_cholesky_size raised to n (one dense-generation launch, batched float64 Cholesky,
      one derivative launch) --
the measurement the round-3 verdict asked for before anyone builds a lock-step batched CG (reference batch mode: gpytorch/kernels/kernel.py:163-208,
test/examples/test_batch_gp_regression.py).  Usage: python scripts/batch_member_timing.py [b+b2] [n+n2]  -> gpurun_out/batch_member_timing.json"""
import json
import os
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpytorch_amd as g  # noqa: E402

bs_list = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "8+64").split("+")]
ns_list = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1000+2000+4000").split("+")]
dev = torch.device("cuda:0")
d = 3


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
    for name, mcs in (("member_loop_bbmm", 0), ("stacked_dense", n)):
        with g.settings.max_cholesky_size(mcs), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ts = []
            for _ in range(3):
                m.zero_grad()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                val = mll(m(X), Y).sum()
                val.backward()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            out[name] = {"ms": [round(1e3 * t, 1) for t in ts], "mll_sum": float(val.detach())}
    out["stacked_over_loop_speedup"] = round(out["member_loop_bbmm"]["ms"][-1] / out["stacked_dense"]["ms"][-1], 2)
    print(out, flush=True)
    return out


results = [run(b, n) for b in bs_list for n in ns_list if b * n * n * 8 * 3 < 1.5e11]
os.makedirs("gpurun_out", exist_ok=True)
json.dump(results, open("gpurun_out/batch_member_timing.json", "w"), indent=1)
