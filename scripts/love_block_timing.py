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


def at_metric_size():
    """n = 500 000 (no dense factor possible): cold posterior at 1000 test points through the model API, LOVE variance against the EXACT-variance
    path of the same model (a 64-column mBCG solve at tolerance 1e-4 on the first 64 test points), per block size / rank."""
    import json as _json
    import time

    import gpytorch_amd as g
    from tests.test_gpu_dense_at_size import synth
    from tests.test_gpu_model import _model

    n, ns, s2 = 500_000, 1000, 0.1
    X, y = synth(n, 3)
    Xs, _ = synth(ns, 3, seed=3)
    S = g.settings
    Xsd = Xs.to(dev)
    rows = []
    _, m, lik = _model("rbf", X, y, 0.25, 1.0, s2, dev, mean=0.0)
    m.eval(), lik.eval()
    with torch.no_grad(), S.max_cholesky_size(0), S.eval_cg_tolerance(1e-4), S.fast_pred_var(False), S.max_preconditioner_size(100), S.max_cg_iterations(4000):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        var_ref = m(Xsd[:64]).variance.double().cpu()
        torch.cuda.synchronize(dev)
        rows.append({"what": "exact variance, 64 test points (64-column solve)", "seconds": time.perf_counter() - t0})
    for blk, rank in ((1, 100), (1, 400), (8, 96), (8, 400), (8, 800), (16, 400), (16, 800), (16, 1600)):
        _, m, lik = _model("rbf", X, y, 0.25, 1.0, s2, dev, mean=0.0)
        m.eval(), lik.eval()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        with torch.no_grad(), S.max_cholesky_size(0), S.eval_cg_tolerance(1e-4), S.fast_pred_var(True), S.max_preconditioner_size(100), \
                S.max_root_decomposition_size(rank), S.max_cg_iterations(4000), S.lanczos_block_size(blk):
            var = m(Xsd).variance.double().cpu()
        torch.cuda.synchronize(dev)
        rows.append({"lanczos_block_size": blk, "love_rank": rank, "seconds_cold_posterior": time.perf_counter() - t0,
                     "var_max_err_over_noise_vs_exact_path": float((var[:64] - var_ref).abs().max() / s2)})
        print(rows[-1], flush=True)
        del m, lik
        torch.cuda.empty_cache()
    with open("gpurun_out/love_block_timing_metric.json", "w") as f:
        _json.dump({"name": "metric", "n": n, "rows": rows}, f, indent=1)


if name == "metric":
    at_metric_size()
    sys.exit(0)
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
