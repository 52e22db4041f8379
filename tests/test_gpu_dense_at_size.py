"""The MLL ingredients AT A BENCHMARKED SIZE against a dense float64 factorisation on the device.

Every parity check the reference holds for this path compares with a dense factorisation
(``test/lazy/test_lazy_evaluated_kernel_tensor.py:84-105``, ``test/distributions/test_multivariate_normal.py:219-237``); the small-shape
tests here do the same up to n = 3000.  An MI355X holds a 1e5 x 1e5 float64 matrix, so BASELINE's C2 (n = 100 000, d = 3, RBF, 64 probes + y)
can be pinned the same way: K_hat is built in float64 with the reference's dense formulas and factorised in place (tests/dense_device.py), then

  * y^T K_hat^-1 y and K_hat^-1 y from the fused path (mBCG at cg_tolerance 1e-4, with and without the pivoted-Cholesky preconditioner)
    must agree with the dense values to the north-star tolerance rtol 1e-3;
  * the stochastic-Lanczos log-determinant (64 probes, fixed generator) must land on log det K_hat within its own sampling error
    (4 standard errors estimated from the per-probe quadratures, plus 1e-3 relative) once the quadrature is converged
    (preconditioner + enough Lanczos steps), and the reference-default setting (20 Lanczos steps) is recorded beside it.

The same for a C3-shaped problem (Matern-5/2, d = 10, rank-100 preconditioner) at n = 60 000 -- with and without the preconditioner both
runs must land on the dense value, which arbitrates between the two tolerance-1 MLL values measured at n = 500 000.
Every run writes its numbers to gpurun_out/dense_at_size_<name>.json (copied into profiles/).
"""
import json
import math
import os
import time

import pytest
import torch

from tests import dense_device as DD

pytestmark = pytest.mark.gpu

LOG_2PI = math.log(2 * math.pi)


def synth(n, d, seed=0):
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(n, d, generator=g, dtype=torch.float32)
    y = torch.sin(2 * math.pi * X[:, 0]) + torch.cos(math.pi * X.sum(-1)) + 0.1 * torch.randn(n, generator=g)
    return X, y


def _per_probe_logdet_samples(t_mats, n):
    from gpytorch_amd.bbmm import slq_logdet

    return torch.stack([slq_logdet(t_mats[j : j + 1], n) for j in range(t_mats.shape[0])])


# One dense float64 factor per model (kind, n, d, lengthscale) for the whole module (round 6: the MLL-ingredient case and the posterior case of a model
# used to factorise the same 80 GB / 29 GB matrix twice -- 9 s / 3 s each); freed by the `dense_factors` fixture when the module is done.
_FACTORS: dict = {}


def _dense_gp(kind, X, y, ls, theta, s2v, dev):
    key = (kind, X.shape[0], X.shape[1], ls, theta, s2v)
    if key not in _FACTORS:
        _FACTORS[key] = DD.DenseGP(kind, X, y, ls, theta, s2v, dev)
    return _FACTORS[key]


@pytest.fixture(scope="module", autouse=True)
def dense_factors():
    yield
    for gp in _FACTORS.values():
        gp.free()
    _FACTORS.clear()
    torch.cuda.empty_cache()


def run_case(name, kind, n, d, ls, dev, probes=64, tight_tol=1e-4, lanczos_steps=(20, 100), ranks=(0, 100), max_iter=6000):
    from gpytorch_amd import backend as B
    from gpytorch_amd import settings
    from gpytorch_amd.bbmm import build_preconditioner, inv_quad_logdet_forward

    theta, s2v = 1.0, 0.1
    X, y = synth(n, d)
    log = {"name": name, "kind": kind, "n": n, "d": d, "lengthscale": ls, "outputscale": theta, "noise": s2v, "probes": probes}
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    gp0 = _dense_gp(kind, X, y, ls, theta, s2v, dev)
    iq, ld, sol = gp0.inv_quad, gp0.logdet, gp0.alpha.reshape(-1)
    torch.cuda.synchronize(dev)
    log["dense"] = {"inv_quad": iq, "logdet": ld, "seconds": time.perf_counter() - t0,
                    "mll_per_datum": -0.5 * (iq + ld + n * LOG_2PI) / n}
    torch.cuda.empty_cache()

    Xd, yd = X.to(dev), y.to(dev)
    xp = B.prep_points(kind, Xd, torch.tensor([ls]), Xd.mean(0))
    sc, s2 = torch.tensor([theta], device=dev), torch.tensor([s2v], device=dev)
    rhs_t = B.to_probe_major(yd.unsqueeze(-1))
    runs = []
    for rank in ranks:
        pre = build_preconditioner(xp, sc, s2, rank=rank, min_size=0) if rank else None
        for steps in lanczos_steps:
            for tol in (tight_tol, 1.0):
                if tol == 1.0 and steps != lanczos_steps[0]:
                    continue
                gen = torch.Generator(device=dev).manual_seed(1234)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                with settings.max_lanczos_quadrature_iterations(steps):
                    res = inv_quad_logdet_forward(xp, sc, s2, rhs_t, num_probes=probes, precond=pre, generator=gen, tolerance=tol, max_iter=max_iter)
                torch.cuda.synchronize(dev)
                sec = time.perf_counter() - t0
                iq_f, ld_f = float(res.inv_quad.sum()), float(res.logdet)
                samples = _per_probe_logdet_samples(res.info.t_mats, n)
                stderr = float(samples.std(unbiased=True)) / math.sqrt(samples.numel())
                xs = res.solves_t[probes, :n].double()
                runs.append({
                    "precond_rank": rank, "lanczos_steps": steps, "cg_tolerance": tol, "cg_iterations": res.info.iterations,
                    "tolerance_reached": bool(res.info.tolerance_reached), "seconds": sec,
                    "inv_quad": iq_f, "inv_quad_rel_err": abs(iq_f - iq) / abs(iq),
                    "solve_rel_err": float((xs - sol).norm() / sol.norm()),
                    "logdet": ld_f, "logdet_rel_err": abs(ld_f - ld) / abs(ld), "logdet_abs_err": ld_f - ld,
                    "logdet_sampling_stderr": stderr,
                    "mll_per_datum": -0.5 * (iq_f + ld_f + n * LOG_2PI) / n,
                })
    log["fused"] = runs
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/dense_at_size_{name}.json", "w") as f:
        json.dump(log, f, indent=1)
    return log


def _check(log, conv_steps):
    """Assertions shared by the cases: tight-tolerance runs vs the dense values."""
    for r in log["fused"]:
        tag = (r["precond_rank"], r["lanczos_steps"], r["cg_tolerance"])
        if r["cg_tolerance"] == 1.0:
            # reference training tolerance: recorded (it is NOT a 1e-3 solve); the value must at least be finite
            assert math.isfinite(r["mll_per_datum"]), tag
            continue
        assert r["tolerance_reached"], tag
        assert r["inv_quad_rel_err"] < 1e-3, (tag, r["inv_quad_rel_err"])
        assert r["solve_rel_err"] < 2e-3, (tag, r["solve_rel_err"])
        if r["precond_rank"] and r["lanczos_steps"] >= conv_steps:
            # converged quadrature: inside the sampling error of the 64-probe estimator
            bound = 4.0 * r["logdet_sampling_stderr"] + 1e-3 * abs(log["dense"]["logdet"])
            assert abs(r["logdet_abs_err"]) < bound, (tag, r["logdet_abs_err"], bound)


def test_c2_mll_ingredients_vs_dense_cholesky(dev):
    """BASELINE C2: RBF, n = 100 000, d = 3, 64 probes + y."""
    log = run_case("c2", "rbf", 100_000, 3, 0.25, dev)
    _check(log, conv_steps=100)


def test_c3_shape_mll_ingredients_vs_dense_cholesky(dev):
    """C3's model (Matern-5/2, d = 10, rank-100 pivoted-Cholesky preconditioner) at n = 60 000: with and without the preconditioner the
    tight-tolerance runs land on the dense float64 values."""
    log = run_case("c3_n60000", "matern52", 60_000, 10, 0.8, dev)
    _check(log, conv_steps=100)


# ---- round 4: the predictive posterior at the same sizes ------------------------------------------------------------------------------
def run_posterior_case(name, kind, n, d, ls, dev, ns=1000,
                       configs=((15, 0.01, False, 100), (15, 0.01, True, 100), (100, 1e-4, False, 100), (100, 1e-4, True, 100), (100, 1e-4, True, 400),
                                (100, 1e-4, True, 1600), (15, 0.01, True, 400, True), (100, 1e-4, False, 100, True))):
    """Predictive mean / variance of f at ``ns`` test points through the model API (``ExactGP.__call__`` in eval mode ->
    ``DefaultPredictionStrategy``: mean cache by mBCG, exact variance by a 1000-column solve, LOVE variance from the Lanczos root) against
    ``K_*X K_hat^-1 y`` and ``diag(K_** - K_*X K_hat^-1 K_X*)`` from the dense float64 factor.  Mirrors
    ``gpytorch/models/exact_prediction_strategies.py:371-478`` and the exact-vs-fast checks of ``test/examples/test_simple_gp_regression.py:386-388,
    396-442`` (LOVE criterion there: |variance error| / noise < 0.05)."""
    import gpytorch_amd as g
    from tests.test_gpu_model import _model

    theta, s2v = 1.0, 0.1
    X, y = synth(n, d)
    Xs, _ = synth(ns, d, seed=3)
    log = {"name": name, "kind": kind, "n": n, "d": d, "test_points": ns, "lengthscale": ls, "outputscale": theta, "noise": s2v}
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    gp = _dense_gp(kind, X, y, ls, theta, s2v, dev)
    mean_ref, var_ref = gp.posterior(Xs)
    mean_ref, var_ref = mean_ref.cpu(), var_ref.cpu()
    torch.cuda.synchronize(dev)
    log["dense_seconds"] = time.perf_counter() - t0
    log["dense"] = {"mean_abs_max": float(mean_ref.abs().max()), "var_min": float(var_ref.min()), "var_median": float(var_ref.median()), "var_max": float(var_ref.max())}
    S = g.settings
    Xsd = Xs.to(dev)
    runs = []
    # (preconditioner rank, eval_cg_tolerance, fast_pred_var, max_root_decomposition_size): the reference defaults first, then a tight solve and
    # LOVE at growing Lanczos rank
    for cfg in configs:
        rank, tol, fast, love_rank = cfg[:4]
        refine = len(cfg) > 4 and cfg[4]          # settings.rhs_refinement: float64 residual replacement for the mean-cache solve
        blk = cfg[5] if len(cfg) > 5 else "auto"  # settings.lanczos_block_size (round 5): rows per Lanczos product of the LOVE cache
        if True:
            if True:
                _, m, lik = _model(kind, X, y, ls, theta, s2v, dev, mean=0.0)   # a fresh model: cold caches
                m.eval()
                lik.eval()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                with torch.no_grad(), S.max_cholesky_size(0), S.eval_cg_tolerance(tol), S.fast_pred_var(fast), S.max_preconditioner_size(rank), \
                        S.max_root_decomposition_size(love_rank), S.max_cg_iterations(6000), S.rhs_refinement(refine), S.lanczos_block_size(blk):
                    pred = m(Xsd)
                    mu, var = pred.mean.double().cpu(), pred.variance.double().cpu()
                torch.cuda.synchronize(dev)
                sec = time.perf_counter() - t0
                runs.append({
                    "precond_rank": rank, "eval_cg_tolerance": tol, "fast_pred_var": fast, "love_rank": love_rank if fast else None, "rhs_refinement": bool(refine), "seconds": sec,
                    "lanczos_block_size": blk if fast else None,
                    "mean_rel_err": float((mu - mean_ref).norm() / mean_ref.norm()),
                    "mean_max_abs_err": float((mu - mean_ref).abs().max()),
                    "var_max_rel_err": float(((var - var_ref).abs() / var_ref).max()),
                    "var_max_err_over_noise": float((var - var_ref).abs().max() / s2v),
                    "yvar_max_rel_err": float(((var - var_ref).abs() / (var_ref + s2v)).max()),
                    # the variance of f itself against rtol 2e-3 + an absolute floor of 2e-6: K_*X is held in float32 (6e-8 relative per entry) and
                    # k_*^T K_hat^-1 k_* sums n products of them whose absolute sum is ~10-30 -> a few 1e-6 is what float32 INPUTS allow whatever the solve
                    "fvar_max_err_over_bound": float(((var - var_ref).abs() / (2e-3 * var_ref + 2e-6)).max()),
                })
                del m, lik
                torch.cuda.empty_cache()
    log["fused"] = runs
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/posterior_at_size_{name}.json", "w") as f:
        json.dump(log, f, indent=1)
    return log


def _check_posterior(log, love_rank_ok=None):
    love = []
    for r in log["fused"]:
        tag = (log["name"], r["precond_rank"], r["eval_cg_tolerance"], r["fast_pred_var"], r["love_rank"])
        if r["eval_cg_tolerance"] <= 1e-4:
            assert r["mean_rel_err"] < 1e-3, (tag, r["mean_rel_err"])
            if not r["fast_pred_var"] and r["rhs_refinement"]:
                # round 5: with settings.rhs_refinement on the n_test-column solve (float64 residual replacement) and a float64 last contraction the variance
                # of F is asserted, not only the variance of y: north_star's "predictive mean/variance rtol 1e-3" at the 2e-3 of the solve path
                assert r["fvar_max_err_over_bound"] < 1.0, (tag, r["fvar_max_err_over_bound"], r["var_max_rel_err"])
            if not r["fast_pred_var"]:
                # the predictive variance of y (what likelihood(model(x)) returns): K_** - K_*X K_hat^-1 K_X* + noise.  The variance of f itself is
                # 1e-4 .. 2e-3 at C2 (1 - 0.9998...): float32 arithmetic cannot hold it to 2e-3 RELATIVE in this or in the reference's own code path
                assert r["yvar_max_rel_err"] < 2e-3, (tag, r["yvar_max_rel_err"])
            else:
                love.append((r["love_rank"], r["var_max_err_over_noise"]))
        else:
            # the reference's default eval tolerance (0.01 on the normalised residual) is not a 1e-3 solve: recorded, bounded loosely; one step of
            # settings.rhs_refinement at the SAME tolerance squares the accuracy of the mean
            assert r["mean_rel_err"] < (1e-3 if r["rhs_refinement"] else 2e-2), (tag, r["mean_rel_err"])
    # LOVE is a rank-k Krylov approximation of K_hat^-1: its error is the ALGORITHM's and falls with the rank (max_root_decomposition_size, reference
    # default 100).  The reference's own criterion -- variance error within 5 % of the noise, test_simple_gp_regression.py:436-442 -- from the
    # rank given (C2: 400; the Matern d = 10 problem does not get there by rank 1600 -- recorded, and asserted to improve monotonically)
    love.sort()
    assert all(b[1] <= max(a[1] * 1.05, 1e-3) for a, b in zip(love, love[1:])), love     # (1e-3 of the noise: the float32 floor of the variance itself)
    if love_rank_ok is not None:
        assert all(err < 0.05 for rank, err in love if rank >= love_rank_ok), love


def test_c2_posterior_vs_dense_cholesky(dev):
    """BASELINE C2 (RBF, n = 100 000, d = 3): 1000 test points, rank-15 (reference default) and rank-100 preconditioner."""
    _check_posterior(run_posterior_case("c2", "rbf", 100_000, 3, 0.25, dev), love_rank_ok=400)


def test_c3_shape_posterior_vs_dense_cholesky(dev):
    """C3's model (Matern-5/2, d = 10) at n = 60 000."""
    _check_posterior(run_posterior_case("c3_n60000", "matern52", 60_000, 10, 0.8, dev))
