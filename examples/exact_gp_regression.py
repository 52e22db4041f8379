#!/usr/bin/env python3
"""Exact GP regression on one MI355X through the gpytorch-shaped API (the workflow of the reference's
examples/02_Scalable_Exact_GPs/KeOps_GP_Regression.ipynb): train hyper-parameters on the BBMM path, then predict with
LOVE variances.

    python examples/exact_gp_regression.py --n 100000 --d 3 --iters 25

Multi-GPU (one process per GPU, probe columns of the MLL and rows of the posterior solves sharded over RCCL):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/exact_gp_regression.py --n 500000
"""
import argparse
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpytorch_amd as gpytorch  # noqa: E402


def synth(n, d, seed):
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(n, d, generator=g)
    y = torch.sin(2 * math.pi * X[:, 0]) + torch.cos(math.pi * X.sum(-1)) + 0.1 * torch.randn(n, generator=g)
    return X, y


class GP(gpytorch.models.ExactGP):
    def __init__(self, x, y, likelihood, ard):
        super().__init__(x, y, likelihood)
        self.mean_module = gpytorch.means.ConstantMean()
        base = gpytorch.kernels.MaternKernel(nu=2.5, ard_num_dims=x.shape[-1] if ard else None,
                                             lengthscale_prior=gpytorch.priors.GammaPrior(3.0, 6.0))
        self.covar_module = gpytorch.kernels.ScaleKernel(base, outputscale_prior=gpytorch.priors.GammaPrior(2.0, 0.15))

    def forward(self, x):
        return gpytorch.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100_000)
    ap.add_argument("--d", type=int, default=3)
    ap.add_argument("--iters", type=int, default=25)
    ap.add_argument("--probes", type=int, default=16)
    ap.add_argument("--ard", action="store_true")
    args = ap.parse_args()

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    group = gpytorch.distributed.init_from_env()  # None on a single GPU
    rank = 0 if group is None else torch.distributed.get_rank(group)

    X, y = synth(args.n, args.d, seed=0)
    Xs, ys = synth(10_000, args.d, seed=3)
    likelihood = gpytorch.likelihoods.GaussianLikelihood().to(dev)
    model = GP(X.to(dev), y.to(dev), likelihood, args.ard).to(dev)
    mll = gpytorch.ExactMarginalLogLikelihood(likelihood, model)
    opt = torch.optim.Adam(model.parameters(), lr=0.1)
    S = gpytorch.settings

    model.train()
    likelihood.train()
    t0 = time.perf_counter()
    with S.max_cholesky_size(0), S.num_trace_samples(args.probes), S.sharding(probe_group=group, row_group=group):
        for it in range(args.iters):
            opt.zero_grad()
            loss = -mll(model(model.train_inputs[0]), model.train_targets)
            loss.backward()
            opt.step()
            if rank == 0 and (it % 5 == 0 or it == args.iters - 1):
                print(f"iter {it:3d}  loss {loss.item():.4f}  lengthscale {model.covar_module.base_kernel.lengthscale.flatten().tolist()}"
                      f"  noise {float(likelihood.noise):.4f}", flush=True)
        torch.cuda.synchronize(dev)
        t_train = time.perf_counter() - t0

        model.eval()
        likelihood.eval()
        t0 = time.perf_counter()
        with torch.no_grad(), S.fast_pred_var():
            pred = likelihood(model(Xs.to(dev)))
            mean, var = pred.mean, pred.variance
        torch.cuda.synchronize(dev)
        t_pred = time.perf_counter() - t0
    if rank == 0:
        rmse = float((mean.cpu() - ys).pow(2).mean().sqrt())
        print(f"train {t_train:.1f} s ({args.iters} iterations), predict {t_pred:.1f} s, test RMSE {rmse:.4f}, "
              f"mean predictive std {float(var.sqrt().mean()):.4f}")
    if group is not None:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
