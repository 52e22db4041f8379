"""MLL forward at the metric shape (n = 500 000, RBF d = 3, 64 probes + y, split contraction) against `settings.max_lanczos_quadrature_iterations` (reference default 20:
it is the FLOOR of the mBCG iteration count whenever a log-determinant is requested) with the rank-256 pivoted-Cholesky preconditioner: how many Lanczos steps
does the stochastic quadrature of log|P^-1 K_hat| need once the preconditioner has clustered the spectrum?  Deviation from a converged evaluation (20 steps, cg_tolerance 1e-3).
    python scripts/mll_quadrature_steps.py -> gpurun_out/mll_quadrature_steps.json"""
import json
import math
import sys
import time

import torch

sys.path.insert(0, ".")
import gpytorch_amd as g  # noqa: E402
from gpytorch_amd import linear_cg as LCG  # noqa: E402
from tests.test_gpu_model import _model  # noqa: E402

dev = torch.device("cuda:0")
S = g.settings
n = 500_000
gen = torch.Generator().manual_seed(0)
X = torch.rand(n, 3, generator=gen)
y = torch.sin(2 * math.pi * X[:, 0]) + torch.cos(math.pi * X.sum(-1)) + 0.1 * torch.randn(n, generator=gen)
_, m, lik = _model("rbf", X, y, 0.25, 1.0, 0.1, dev, mean=0.0)
m.train(), lik.train()


def iql(rank, tol, steps, seed=5):
    torch.manual_seed(seed)
    with torch.no_grad(), S.max_cholesky_size(0), S.num_trace_samples(64), S.max_preconditioner_size(rank), S.cg_tolerance(tol), S.max_lanczos_quadrature_iterations(steps):
        mvn = lik(m(m.train_inputs[0]))
        op = mvn.lazy_covariance_matrix.evaluate_kernel()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        iq, ld = op.inv_quad_logdet((m.train_targets - mvn.mean).unsqueeze(-1), logdet=True)
        torch.cuda.synchronize()
        return float(iq), float(ld), LCG.LAST_INFO.iterations, time.perf_counter() - t0


iql(256, 1.0, 20)
iq0, ld0, it0, s0 = iql(256, 1e-3, 20)
rows = [{"what": "converged reference", "rank": 256, "cg_tolerance": 1e-3, "quadrature_steps": 20, "inv_quad": iq0, "logdet": ld0, "cg_iterations": it0, "seconds": s0}]
for rank in (256, 100):
    for steps in (20, 15, 12, 10, 8, 6):
        iq, ld, it, sec = iql(rank, 1.0, steps)
        rows.append({"rank": rank, "cg_tolerance": 1.0, "quadrature_steps": steps, "cg_iterations": it, "seconds": sec,
                     "inv_quad_rel_dev": abs(iq - iq0) / abs(iq0), "logdet_rel_dev": abs(ld - ld0) / abs(ld0),
                     "mll_per_datum_abs_dev": abs((iq + ld) - (iq0 + ld0)) / (2 * n)})
        print(rows[-1], flush=True)
json.dump({"n": n, "rows": rows}, open("gpurun_out/mll_quadrature_steps.json", "w"), indent=1)
