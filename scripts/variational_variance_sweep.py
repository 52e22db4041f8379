"""Exact predictive variance with settings.rhs_refinement (round 6: bbmm.variational_inv_quad -- X^T (2 B - K_hat X), one float64 product, no second solve)
at C2 (n = 100 000, 1000 test points) against the dense float64 factor, per eval_cg_tolerance: the form is second order in the ENERGY norm of the solve
error, which a residual tolerance of 1e-4 does not make small at kappa ~ 1e5 -- how much tighter must the float32 solve stop, and what does it cost?
    python scripts/variational_variance_sweep.py -> gpurun_out/posterior_at_size_c2_variational.json"""
import json
import sys

import torch

sys.path.insert(0, ".")
from tests.test_gpu_dense_at_size import run_posterior_case  # noqa: E402

dev = torch.device("cuda:0")
cfgs = [(100, 1e-4, False, 100, False)]
for rank in (100, "auto"):
    for tol in (1e-4, 3e-5, 1e-5):     # (float32 mBCG does not reach 1e-6: the recurrence stalls for thousands of iterations)
        cfgs.append((rank, tol, False, 100, True))
log = run_posterior_case("c2_variational", "rbf", 100_000, 3, 0.25, dev, configs=tuple(cfgs))
for r in log["fused"]:
    print({k: r[k] for k in ("precond_rank", "eval_cg_tolerance", "rhs_refinement", "seconds", "mean_rel_err", "var_max_err_over_noise", "fvar_max_err_over_bound")}, flush=True)
