"""Where the cold prediction of the road3d-shaped workload spends its time (217 437 training and test points, the hyper-parameters the 25 Adam iterations
end at): wall-clock of the mean (mean-cache solve + K_*X alpha) and of the fast_pred_var variance (LOVE root + K_*X R), un-culled, on the library
defaults.  python scripts/road3d_predict_profile.py -> one JSON line; under rocprofv3 --kernel-trace --stats the kernel time beside it."""
import json
import sys
import time
import warnings

import torch

import os  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import gpytorch_amd as g  # noqa: E402
from reference_workloads import _model, road_like  # noqa: E402

dev = torch.device("cuda:0")
n = 217_437
Xall, yall = road_like(2 * n, 0)
X, y, Xs = Xall[:n], yall[:n], Xall[n:]
m, lik = _model(g, "matern52", X, y, dev)
m.covar_module.base_kernel.lengthscale, m.covar_module.outputscale, lik.noise = 0.354, 0.149, 0.074
m.eval(), lik.eval()
Xsd = Xs.to(dev)
rec = {}
warnings.simplefilter("ignore")
with torch.no_grad(), g.settings.fast_pred_var():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = m(Xsd)
    mu = out.mean
    torch.cuda.synchronize()
    rec["latent_mean_seconds"] = time.perf_counter() - t0
    from gpytorch_amd import linear_cg as LCG

    rec["mean_cache_cg_iterations"] = LCG.LAST_INFO.iterations
    t0 = time.perf_counter()
    var = out.variance
    torch.cuda.synchronize()
    rec["latent_variance_seconds"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    pred = lik(out)
    _ = pred.mean, pred.variance
    torch.cuda.synchronize()
    rec["likelihood_seconds"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    out2 = m(Xsd)
    _ = out2.mean, out2.variance
    torch.cuda.synchronize()
    rec["warm_seconds"] = time.perf_counter() - t0
print(json.dumps(rec))
