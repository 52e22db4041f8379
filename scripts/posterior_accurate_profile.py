"""The cold posterior at the metric shape with the settings that meet BASELINE's tolerance (bench.py extras, third posterior row): wall time of
the first call, for rocprofv3 --kernel-trace --stats.  python scripts/posterior_accurate_profile.py [block] [rank]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import gpytorch_amd as g  # noqa: E402
from tests.test_gpu_dense_at_size import synth  # noqa: E402
from tests.test_gpu_model import _model  # noqa: E402

dev = torch.device("cuda:0")
blk = sys.argv[1] if len(sys.argv) > 1 else "auto"
blk = blk if blk == "auto" else int(blk)
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 400
n, ns, s2 = 500_000, 10_000, 0.1
X, y = synth(n, 3)
Xs, _ = synth(ns, 3, seed=3)
S = g.settings
Xsd = Xs.to(dev)
for rep in range(2):
    _, m, lik = _model("rbf", X, y, 0.25, 1.0, s2, dev, mean=0.0)
    m.eval(), lik.eval()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    with torch.no_grad(), S.max_cholesky_size(0), S.eval_cg_tolerance(1e-4), S.fast_pred_var(True), S.max_preconditioner_size("auto"), \
            S.max_root_decomposition_size(rank), S.max_cg_iterations(4000), S.lanczos_block_size(blk):
        out = m(Xsd)
        var = out.variance
        mean = out.mean
    torch.cuda.synchronize(dev)
    print({"rep": rep, "block": blk, "rank": rank, "cold_seconds": time.perf_counter() - t0, "mean_abs_max": float(mean.abs().max()), "var_range": [float(var.min()), float(var.max())]}, flush=True)
