"""LOVE cache at size: single-vector Lanczos (the reference's recurrence) against block Lanczos (settings.lanczos_block_size) --
cold posterior wall time (mean cache + LOVE cache + 1000 test points through the model API) and the variance error against the
dense float64 factor, in units of the noise (the reference's criterion: < 0.05, test_simple_gp_regression.py:436-442).

    python scripts/love_block_timing.py [c2|c3shape] -> gpurun_out/love_block_timing_<name>.json
"""
import json
import sys

import torch

sys.path.insert(0, ".")
from tests.test_gpu_dense_at_size import run_posterior_case  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
dev = torch.device("cuda:0")
# (preconditioner rank, eval_cg_tolerance, fast_pred_var, LOVE rank, rhs_refinement, block size)
cfgs = []
for blk, ranks in ((1, (100, 400)), (8, (96, 400, 800, 1600)), (16, (400, 800, 1600))):
    for r in ranks:
        cfgs.append((100, 1e-4, True, r, False, blk))
if name == "c2":
    log = run_posterior_case("c2_love_block", "rbf", 100_000, 3, 0.25, dev, configs=tuple(cfgs))
else:
    log = run_posterior_case("c3shape_love_block", "matern52", 60_000, 10, 0.8, dev, configs=tuple(cfgs))
rows = [{k: r[k] for k in ("lanczos_block_size", "love_rank", "seconds", "var_max_err_over_noise", "mean_rel_err")} for r in log["fused"]]
with open(f"gpurun_out/love_block_timing_{name}.json", "w") as f:
    json.dump({"name": name, "n": log["n"], "rows": rows}, f, indent=1)
for r in rows:
    print(r)
