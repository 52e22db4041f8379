"""The reference's own published workloads for this path, end to end through the gpytorch-shaped API, on synthetic data of the same shape
(no network: the UCI files cannot be fetched) -- `python bench.py --config road3d | protein` dispatches here.

  road3d   examples/02_Scalable_Exact_GPs/KeOps_GP_Regression.ipynb: n = 217 437 training points, d = 3, standardised inputs and targets,
           ScaleKernel(Matern-5/2) + ConstantMean + GaussianLikelihood, lengthscale initialised to 0.05, 25 Adam iterations (lr 0.1) on the library
           defaults (num_trace_samples 10, cg_tolerance 1, rank-15 pivoted-Cholesky preconditioner), then `fast_pred_var` prediction on 217 437 test
           points.  The notebook reports "a matter of minutes" for training on one GPU and RMSE 0.138 (cells 7-9).
  protein  examples/02_Scalable_Exact_GPs/Simple_MultiGPU_GP_Regression.ipynb: n = 36 584, d = 9, ScaleKernel(RBF), `max_preconditioner_size(100)`,
           full-batch L-BFGS (the notebook's run converged after 5 iterations), caches computed on two test points, then 9 146 test points with
           warm caches under `fast_pred_var`: 1.88 s wall on the notebook's 2-GPU box.

Synthetic stand-ins: inputs N(0, 1)^d clipped to +-3 (the notebooks z-score every feature), targets a smooth function + noise, z-scored.  The
3droad stand-in puts the points along random smooth curves in the plane with a slowly varying third coordinate (a road network is locally
one-dimensional; a uniform cloud at lengthscale 0.05 would make K nearly diagonal and say nothing about the kernels).
Output: one JSON line (rank 0): seconds per training iteration (all of them listed), total, prediction time, which kernel path ran."""
from __future__ import annotations

import json
import math
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def road_like(n, seed=0):
    """n points along ~400 random smooth planar curves + a slowly varying third coordinate; z-scored.  Returns (X [n, 3], y [n])."""
    g = torch.Generator().manual_seed(seed)
    roads = 400
    per = (n + roads - 1) // roads
    t = torch.linspace(0, 1, per).unsqueeze(0)                                   # [1, per]
    p0 = torch.rand(roads, 1, 2, generator=g) * 10.0
    ang = torch.rand(roads, 1, generator=g) * 2 * math.pi
    curv = (torch.rand(roads, 1, generator=g) - 0.5) * 6.0
    length = 0.5 + 2.5 * torch.rand(roads, 1, generator=g)
    th = ang + curv * t                                                           # heading along the road
    step = length / per
    xy = p0 + torch.stack([torch.cumsum(torch.cos(th) * step, 1), torch.cumsum(torch.sin(th) * step, 1)], -1)
    xy = xy.reshape(-1, 2)[:n]
    xy = xy + 0.002 * torch.randn(xy.shape, generator=g)
    third = torch.sin(0.7 * xy[:, 0]) * torch.cos(0.5 * xy[:, 1]) + 0.05 * torch.randn(n, generator=g)
    X = torch.cat([xy, third.unsqueeze(-1)], -1)
    alt = torch.sin(0.9 * xy[:, 0] + 0.3) + 0.6 * torch.cos(1.3 * xy[:, 1]) + 0.3 * torch.sin(2.1 * xy[:, 0] * 0.5 + xy[:, 1])
    y = alt + 0.1 * torch.randn(n, generator=g)
    perm = torch.randperm(n, generator=g)
    X, y = X[perm], y[perm]
    X = (X - X.mean(0, keepdim=True)) / (X.std(0, keepdim=True) + 1e-6)
    return X.contiguous(), ((y - y.mean()) / y.std()).contiguous()


def gaussian_features(n, d, seed=0):
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(n, d, generator=g).clamp_(-3.0, 3.0)
    w = torch.randn(d, generator=g) / math.sqrt(d)
    y = torch.sin(X @ w * 1.5) + 0.5 * torch.cos(X[:, 0] * X[:, 1]) + 0.3 * torch.randn(n, generator=g)
    X = (X - X.mean(0, keepdim=True)) / (X.std(0, keepdim=True) + 1e-6)
    return X.contiguous(), ((y - y.mean()) / y.std()).contiguous()


def _model(g, kind, X, y, dev):
    class ExactGPModel(g.models.ExactGP):
        def __init__(self, x, yy, lik):
            super().__init__(x, yy, lik)
            self.mean_module = g.means.ConstantMean()
            base = g.kernels.MaternKernel(nu=2.5) if kind == "matern52" else g.kernels.RBFKernel()
            self.covar_module = g.kernels.ScaleKernel(base)

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    lik = g.likelihoods.GaussianLikelihood().to(dev)
    return ExactGPModel(X.to(dev), y.to(dev), lik).to(dev), lik


def _path(m, lik):
    """Which generation kernel the training covariance takes at the CURRENT hyper-parameters (backend.gram_mode)."""
    from gpytorch_amd import backend as B

    with torch.no_grad():
        op = lik(m(m.train_inputs[0])).lazy_covariance_matrix
        p1, _ = op.kernel_op.prepared()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mode = B.gram_mode(p1, p1)
        wide = 0 if mode != 2 else p1.n - p1.sorted_view().n_block   # rows on the direct-difference kernels (medium rows -- 128-row blocks -- stay on the Gram form)
        _path.last_regions = None if mode != 2 else {"compact_rows": p1.sorted_view().n_compact, "medium_rows": p1.sorted_view().n_block - p1.sorted_view().n_compact,
                                                      "wide_rows": wide}
    return {0: "direct-difference kernels (outside the Gram-form policy)", 1: "Gram form, cloud-centred", 2: "Gram form, block-centred"}[mode], float(p1.zmax2), wide


def road3d(dev, n=217_437, n_test=None, iters=25, seed=0, far_eps=None):
    """far_eps: settings.far_pair_cutoff for the whole run (None = the library default: every pair)."""
    import contextlib

    import gpytorch_amd as g

    n_test = n if n_test is None else n_test
    Xall, yall = road_like(n + n_test, seed)
    X, y, Xs, ys = Xall[:n], yall[:n], Xall[n:], yall[n:]
    m, lik = _model(g, "matern52", X, y, dev)
    m.covar_module.base_kernel.lengthscale = 0.05                               # notebook cell 6
    m.train()
    lik.train()
    opt = torch.optim.Adam(m.parameters(), lr=0.1)
    mll = g.ExactMarginalLogLikelihood(lik, m)
    from gpytorch_amd import linear_cg as LCG

    secs, losses, cg_its, paths, ls_hist = [], [], [], [], []
    with warnings.catch_warnings(), (g.settings.far_pair_cutoff(far_eps) if far_eps else contextlib.nullcontext()):
        warnings.simplefilter("ignore")
        for i in range(iters):
            if i in (0, iters - 1):
                paths.append(_path(m, lik))
                m.train(), lik.train()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            opt.zero_grad()
            loss = -mll(m(m.train_inputs[0]), m.train_targets)
            it = LCG.LAST_INFO.iterations
            loss.backward()
            opt.step()
            torch.cuda.synchronize(dev)
            secs.append(time.perf_counter() - t0)
            losses.append(float(loss.detach()))
            cg_its.append(it)
            ls_hist.append(float(m.covar_module.base_kernel.lengthscale.detach().reshape(-1)[0]))
        m.eval()
        lik.eval()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        with torch.no_grad(), g.settings.fast_pred_var():
            pred = lik(m(Xs.to(dev)))
            mu, var = pred.mean, pred.variance
        torch.cuda.synchronize(dev)
        pred_s = time.perf_counter() - t0
    rmse = float((mu.cpu() - ys).square().mean().sqrt())
    return {
        "workload": "road3d-shaped (KeOps_GP_Regression.ipynb): synthetic road-like cloud, Matern-5/2, 25 Adam iterations, library defaults, then fast_pred_var prediction",
        "n": n, "d": 3, "n_test": n_test, "iterations": iters,
        "seconds_per_iteration": secs, "seconds_per_iteration_median": sorted(secs)[len(secs) // 2], "training_seconds": sum(secs),
        "cg_iterations": cg_its, "loss_first_last": [losses[0], losses[-1]],
        "kernel_path_first_last": [p[0] for p in paths], "max_sq_scaled_radius_first_last": [p[1] for p in paths], "wide_rows_first_last": [p[2] for p in paths],
        "hyper": {"lengthscale": float(m.covar_module.base_kernel.lengthscale.detach().reshape(-1)[0]), "outputscale": float(m.covar_module.outputscale.detach()),
                  "noise": float(lik.noise.detach().reshape(-1)[0])},
        "far_pair_cutoff": far_eps, "lengthscale_per_iteration": ls_hist,
        "prediction_seconds_cold_caches": pred_s, "test_rmse": rmse, "variance_min": float(var.min()),
        "reference": "notebook: 'a matter of minutes' for 25 iterations on one GPU (KeOps), RMSE 0.138 on the real data",
    }


def protein(dev, n=36_584, n_test=9_146, d=9, lbfgs_iters=5, seed=0):
    import gpytorch_amd as g

    Xall, yall = gaussian_features(n + n_test, d, seed)
    X, y, Xs, ys = Xall[:n], yall[:n], Xall[n:], yall[n:]
    m, lik = _model(g, "rbf", X, y, dev)
    m.train()
    lik.train()
    mll = g.ExactMarginalLogLikelihood(lik, m)
    opt = torch.optim.LBFGS(m.parameters(), lr=0.1, max_iter=10, line_search_fn="strong_wolfe")   # notebook: FullBatchLBFGS(lr=0.1), max_ls 10
    evals, secs, losses = [0], [], []
    S = g.settings
    with warnings.catch_warnings(), S.max_preconditioner_size(100), S.deterministic_probes(True):   # one probe draw: a deterministic objective for the line search
        warnings.simplefilter("ignore")

        def closure():
            opt.zero_grad()
            loss = -mll(m(m.train_inputs[0]), m.train_targets)
            loss.backward()
            evals[0] += 1
            return loss

        for _ in range(lbfgs_iters):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            loss = opt.step(closure)
            torch.cuda.synchronize(dev)
            secs.append(time.perf_counter() - t0)
            losses.append(float(loss.detach()))
        S.deterministic_probes.reset()
        path = _path(m, lik)
        m.eval()
        lik.eval()
        Xsd = Xs.to(dev)
        with torch.no_grad(), S.fast_pred_var():
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            _ = m(Xsd[:2])                                                         # notebook: compute the test-time caches on two points
            torch.cuda.synchronize(dev)
            cache_s = time.perf_counter() - t0
            t0 = time.perf_counter()
            latent = m(Xsd)
            mu, var = latent.mean, latent.variance
            torch.cuda.synchronize(dev)
            pred_s = time.perf_counter() - t0
    return {
        "workload": "protein-shaped (Simple_MultiGPU_GP_Regression.ipynb): synthetic, RBF, max_preconditioner_size(100), L-BFGS, warm-cache fast_pred_var prediction",
        "n": n, "d": d, "n_test": n_test, "lbfgs_iterations": lbfgs_iters, "closure_evaluations": evals[0],
        "seconds_per_lbfgs_iteration": secs, "training_seconds": sum(secs), "seconds_per_closure_evaluation": sum(secs) / max(evals[0], 1),
        "loss_per_iteration": losses, "kernel_path": path[0],
        "hyper": {"lengthscale": float(m.covar_module.base_kernel.lengthscale.detach().reshape(-1)[0]), "noise": float(lik.noise.detach().reshape(-1)[0])},
        "cache_seconds": cache_s, "prediction_seconds_warm_caches": pred_s, "test_rmse": float((mu.cpu() - ys).square().mean().sqrt()),
        "reference": "notebook (2 GPUs, MultiDeviceKernel): 9 146 test points in 1.88 s wall with warm caches; training converged in 5 L-BFGS iterations",
    }


def main(config, gpus=1, size=None, steps=None, far_eps=None):
    assert gpus == 1, "the reference workloads are single-process runs (probe sharding of the training MLL: settings.sharding)"
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if config == "road3d":
        kw = {} if size is None else {"n": size, "n_test": size}
        rec = road3d(dev, iters=steps or 25, far_eps=far_eps, **kw)
        value, metric = rec["seconds_per_iteration_median"], "road3d_training_iteration_seconds"
    else:
        kw = {} if size is None else {"n": size, "n_test": max(2, size // 4)}
        rec = protein(dev, lbfgs_iters=steps or 5, **kw)
        value, metric = rec["prediction_seconds_warm_caches"], "protein_prediction_seconds"
    line = {"metric": metric, "value": value, "unit": "s", "n_gpus": 1, "higher_is_better": False, "data": "synthetic", "dtype": "f32",
            "vs_baseline": None, "config": rec}
    print(json.dumps(line), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], size=int(sys.argv[2]) if len(sys.argv) > 2 else None, steps=int(sys.argv[3]) if len(sys.argv) > 3 else None))
