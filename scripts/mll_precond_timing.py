"""The wall-clock half of BASELINE's metric at the metric shape (n = 500 000, d = 3, RBF, 64 probes + y) through the model API on the library
defaults (split contraction): ExactMarginalLogLikelihood forward + backward per pivoted-Cholesky preconditioner rank
(`settings.max_preconditioner_size`, re-exported by the reference at gpytorch/settings.py:6-31), with the CG iteration count and the deviation of
y^T K^-1 y / log|K| from a tight run, then the cold posterior per (preconditioner rank, eval_cg_tolerance, LOVE rank / block).

    python scripts/mll_precond_timing.py [mll|posterior|both] [n] -> gpurun_out/mll_precond_timing_n<n>.json
"""
import json
import math
import sys
import time

import torch

sys.path.insert(0, ".")
import gpytorch_amd as g  # noqa: E402
from gpytorch_amd import linear_cg as LCG  # noqa: E402
from tests.test_gpu_model import _model  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "both"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
dev = torch.device("cuda:0")
S = g.settings
t_probes, s2 = 64, 0.1


def synth(n_, d, seed=0):
    gen = torch.Generator().manual_seed(seed)
    X = torch.rand(n_, d, generator=gen, dtype=torch.float32)
    y = torch.sin(2 * math.pi * X[:, 0]) + torch.cos(math.pi * X.sum(-1)) + 0.1 * torch.randn(n_, generator=gen)
    return X, y


def sync():
    torch.cuda.synchronize(dev)


X, y = synth(n, 3)
out = {"n": n, "d": 3, "kind": "rbf", "probes": t_probes, "contraction": "library default (split)"}


def iql(m, lik, rank, tol, probes=t_probes):
    """(inv_quad, logdet, iterations, seconds) of one no-grad evaluation of the MLL's inv_quad_logdet."""
    m.train(), lik.train()
    with torch.no_grad(), S.max_cholesky_size(0), S.num_trace_samples(probes), S.max_preconditioner_size(rank), S.cg_tolerance(tol), S.max_cg_iterations(4000):
        mvn = lik(m(m.train_inputs[0]))
        op = mvn.lazy_covariance_matrix.evaluate_kernel()
        sync()
        t0 = time.perf_counter()
        iq, ld = op.inv_quad_logdet((m.train_targets - mvn.mean).unsqueeze(-1), logdet=True)
        sync()
        return float(iq), float(ld), LCG.LAST_INFO.iterations, time.perf_counter() - t0


if what in ("mll", "both"):
    _, m, lik = _model("rbf", X, y, 0.25, 1.0, s2, dev, mean=0.0)
    mll = g.ExactMarginalLogLikelihood(lik, m)
    torch.manual_seed(11)
    iq_ref, ld_ref, it_ref, sec_ref = iql(m, lik, 100, 1e-3)
    out["tight_reference"] = {"what": "rank-100 preconditioner, cg_tolerance 1e-3, 64 probes", "inv_quad": iq_ref, "logdet": ld_ref, "cg_iterations": it_ref, "seconds": sec_ref}
    print(out["tight_reference"], flush=True)
    rows = []
    for rank in (0, 15, 100, 128, 256):   # (gpamd_pivoted_cholesky_f32 / the fused preconditioner apply: rank <= 512 since round 6)
        m.train(), lik.train()
        rec = {"max_preconditioner_size": rank}
        with S.max_cholesky_size(0), S.num_trace_samples(t_probes), S.max_preconditioner_size(rank):
            for rep in range(3):
                for p in m.parameters():
                    p.grad = None
                sync()
                t0 = time.perf_counter()
                loss = -mll(m(m.train_inputs[0]), m.train_targets)
                sync()
                t1 = time.perf_counter()
                loss.backward()
                sync()
                t2 = time.perf_counter()
                rec.setdefault("forward_ms_all", []).append((t1 - t0) * 1e3)
                rec.setdefault("backward_ms_all", []).append((t2 - t1) * 1e3)
                rec.setdefault("cg_iterations_all", []).append(LCG.LAST_INFO.iterations)
                rec.setdefault("mll_all", []).append(float(-loss))
            rec["forward_ms"] = min(rec["forward_ms_all"][1:])
            rec["backward_ms"] = min(rec["backward_ms_all"][1:])
            rec["grad_raw_lengthscale"] = float(m.covar_module.base_kernel.raw_lengthscale.grad.sum())
            rec["grad_raw_noise"] = float(lik.noise_covar.raw_noise.grad.sum())
        iq, ld, it, sec = iql(m, lik, rank, 1.0)
        rec.update(inv_quad=iq, logdet=ld, inv_quad_rel_dev=abs(iq - iq_ref) / abs(iq_ref), logdet_rel_dev=abs(ld - ld_ref) / abs(ld_ref),
                   iql_seconds=sec, iql_cg_iterations=it)
        # preconditioner build alone
        if rank:
            from gpytorch_amd import backend as B
            from gpytorch_amd.bbmm import build_preconditioner

            xp = B.prep_points("rbf", X.to(dev), torch.tensor([0.25], device=dev), X.to(dev).mean(0))
            sc, nz = torch.tensor([1.0], device=dev), torch.tensor([s2], device=dev)
            build_preconditioner(xp, sc, nz, rank=rank, min_size=0)
            sync()
            t0 = time.perf_counter()
            build_preconditioner(xp, sc, nz, rank=rank, min_size=0)
            sync()
            rec["preconditioner_build_ms"] = (time.perf_counter() - t0) * 1e3
        rows.append(rec)
        print(rec, flush=True)
    out["mll"] = rows
    del m, lik, mll
    torch.cuda.empty_cache()

if what in ("posterior", "both"):
    ns = 1000
    Xs, _ = synth(ns, 3, seed=3)
    Xsd = Xs.to(dev)
    _, m, lik = _model("rbf", X, y, 0.25, 1.0, s2, dev, mean=0.0)
    m.eval(), lik.eval()
    # references: tight mean (tolerance 1e-6 + one float64 refinement step) on all test points, exact variance on the first 64
    with torch.no_grad(), S.max_cholesky_size(0), S.eval_cg_tolerance(1e-6), S.skip_posterior_variances(), S.max_preconditioner_size(100), S.max_cg_iterations(4000), S.rhs_refinement():
        mean_ref = m(Xsd).mean.double().cpu()
    m.train(), m.eval()
    with torch.no_grad(), S.max_cholesky_size(0), S.eval_cg_tolerance(1e-4), S.fast_pred_var(False), S.max_preconditioner_size(100), S.max_cg_iterations(4000):
        var_ref = m(Xsd[:64]).variance.double().cpu()
    rows = []
    cases = [
        # (preconditioner rank, eval_cg_tolerance, LOVE rank, block)
        (15, 1e-2, 100, 1), (100, 1e-2, 100, 1), (100, 1e-4, 400, 16), (128, 1e-4, 400, 16),
        (192, 1e-4, 400, 16), (256, 1e-4, 400, 16), (384, 1e-4, 400, 16), (256, 1e-3, 400, 16), (256, 1e-4, 384, 32),
    ]
    for rank, tol, love, blk in cases:
        best = None
        for rep in range(2):
            m.train(), m.eval()
            sync()
            t0 = time.perf_counter()
            with torch.no_grad(), S.max_cholesky_size(0), S.eval_cg_tolerance(tol), S.fast_pred_var(True), S.max_preconditioner_size(rank), \
                    S.max_root_decomposition_size(love), S.max_cg_iterations(4000), S.lanczos_block_size(blk):
                pred = m(Xsd)
                mu, var = pred.mean.double().cpu(), pred.variance.double().cpu()
            sync()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        rec = {"max_preconditioner_size": rank, "eval_cg_tolerance": tol, "love_rank": love, "lanczos_block_size": blk, "seconds_cold_posterior": best,
               "mean_cache_cg_iterations": LCG.LAST_INFO.iterations,
               "mean_rel_err_vs_tight": float((mu - mean_ref).abs().max() / mean_ref.abs().max()),
               "var_max_err_over_noise_vs_exact_path": float((var[:64] - var_ref).abs().max() / s2)}
        rows.append(rec)
        print(rec, flush=True)
    out["posterior"] = rows

with open(f"gpurun_out/mll_precond_timing_n{n}.json", "w") as f:
    json.dump(out, f, indent=1)
