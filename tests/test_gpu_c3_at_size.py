"""BASELINE C3 at ITS OWN size (Matern-5/2, n = 500 000, d = 10, rank-100 pivoted-Cholesky preconditioner): the marginal-log-likelihood
ingredients of the fused float32 path against the SAME algorithm in float64 -- fused float64 products (``csrc/kv_f64.hpp``), the float64
instantiation of the device-resident mBCG (``gpamd_cg64_*``), the same probe vectors, the same preconditioner factor, the same stopping rule.

No dense factor exists at this size (2 TB in float64); what CAN be pinned is that float32 arithmetic -- hi/lo-split Gram expansion, split
contraction, float32 mBCG recurrences, split W contraction of the backward -- changes nothing an exact-arithmetic run of the reference's
algorithm would produce: y^T K_hat^-1 y, the solve of the y column, the stochastic-Lanczos quadrature of the fixed probes and the three
hyper-parameter gradients (``test/lazy/test_lazy_evaluated_kernel_tensor.py:84-105`` is the reference's pattern: value and gradients against a
higher-precision evaluation of the same quantity).  The dense-truth half of the argument is ``tests/test_gpu_dense_at_size.py`` /
``tests/test_gpu_grad_at_size.py`` at n = 60 000 of the same model.  Numbers -> gpurun_out/c3_at_size_vs_float64.json."""
import json
import os
import time

import pytest
import torch

from tests.test_gpu_parity_at_size import synth

pytestmark = pytest.mark.gpu


def _grads(xp32, res, ls, os_, dev):
    """(d/d lengthscale, d/d outputscale, d/d noise) of inv_quad + logdet from one forward result: the A.6 backward.  The left / right vectors are the
    run's own (float32 or float64 solves); the bilinear derivative sum_c left_c^T (dK) right_c is evaluated by the product's fused kernel on the float32
    prepared points in BOTH cases -- that kernel is pinned on its own at this very size against float64 block sums (tests/test_gpu_grad_at_size.py:
    <= 3e-7 of sum |terms| per 128-row block at n = 500 000), and the float64 row-block derivative would cost 3.5 minutes here.  What this comparison
    isolates is what float32 SOLVES do to the gradient; the noise derivative (a plain inner product) is taken in the run's own precision."""
    from gpytorch_amd import backend as B
    from gpytorch_amd.bbmm import backward_vectors
    from gpytorch_amd.functions import hyper_grads

    wd = res.solves_t.dtype
    one = torch.ones((), device=dev, dtype=wd)
    t = res.zt.shape[0]
    left, right, _ = backward_vectors(res, torch.ones(1, device=dev, dtype=wd), one, t)
    parts = []
    for sl in (slice(0, t), slice(t, None)):          # the log-determinant (probe) part and the data-fit (y) part of every derivative
        lp, rp = left[sl].contiguous(), right[sl].contiguous()
        d_ls, d_os = hyper_grads(xp32, xp32, ls.float(), os_.float(), lp.float(), rp.float())
        parts.append((float(d_ls.sum()), float(d_os.sum()), float(B.coldot(lp, rp, xp32.n).sum())))
    total = tuple(p + q for p, q in zip(*parts))
    return total, parts[0], parts[1]


def _run_both_precisions(dev, n, tol, max_iter):
    """The C3 model's MLL ingredients and gradients on the fused float32 path and on the float64 BBMM path (same probes, same preconditioner factor)."""
    from gpytorch_amd import backend as B
    from gpytorch_amd.bbmm import build_preconditioner, inv_quad_logdet_forward, preconditioner_from_factor

    kind, d, ls, probes = "matern52", 10, 0.8, 3   # (3 probes + y = 4 columns: the float64 run takes the VALU-contraction kernel, kv_f64v)
    X, y = synth(n, d)
    Xd, yd = X.to(dev), y.to(dev)
    lsv = torch.tensor([ls], device=dev)
    log = {"kind": kind, "n": n, "d": d, "probes": probes, "cg_tolerance": tol, "preconditioner_rank": 100}
    runs = {}
    pre32 = None
    zprobe = None
    xp32 = None
    for wd in (torch.float32, torch.float64):
        xp = B.prep_points(kind, Xd.to(wd), lsv.to(wd), Xd.mean(0).to(wd))
        xp32 = xp if wd == torch.float32 else xp32
        sc, s2 = torch.ones(1, device=dev, dtype=wd), torch.full((1,), 0.1, device=dev, dtype=wd)
        rhs_t = B.to_probe_major(yd.unsqueeze(-1), wd)
        if wd == torch.float32:
            assert xp.fused
            pre = pre32 = build_preconditioner(xp, sc, s2, rank=100, min_size=0)
            kw = {"generator": torch.Generator(device=dev).manual_seed(11), "num_probes": probes}
        else:
            assert B.fused_f64(xp, xp)
            pre = preconditioner_from_factor(pre32.lt, n, s2, wd)          # the SAME factor L: P, Q1 and log|P| re-derived in float64
            kw = {"probes": zprobe}                                         # the SAME probe vectors (un-normalised [n, t])
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        res = inv_quad_logdet_forward(xp, sc, s2, rhs_t, precond=pre, tolerance=tol, max_iter=max_iter, **kw)
        torch.cuda.synchronize(dev)
        fwd = time.perf_counter() - t0
        assert res.info.tolerance_reached
        if wd == torch.float32:
            zprobe = (res.zt[:, :n] * res.znorm.unsqueeze(-1)).t().to(torch.float64)
        t0 = time.perf_counter()
        g, g_logdet, g_data = _grads(xp32, res, lsv, sc, dev)
        torch.cuda.synchronize(dev)
        runs[wd] = {"iterations": res.info.iterations, "inv_quad": float(res.inv_quad.sum()), "logdet": float(res.logdet), "logdet_slq_part": float(res.logdet_pinvk),
                    "logdet_precond_part": float(pre.logdet), "grad_lengthscale_outputscale_noise": g, "grad_logdet_part": g_logdet, "grad_data_fit_part": g_data,
                    "forward_seconds": fwd, "backward_seconds": time.perf_counter() - t0,
                    "ysol": res.solves_t[probes, :n].double()}
        del res
        if wd == torch.float64:
            del xp
        torch.cuda.empty_cache()
    a, b = runs[torch.float32], runs[torch.float64]
    ysol_err = float((a["ysol"] - b["ysol"]).norm() / b["ysol"].norm())
    for r in (a, b):
        r.pop("ysol")
    log["float32_fused"], log["float64_bbmm"] = a, b
    log["y_solve_rel_l2_difference"] = ysol_err
    return log, a, b, ysol_err


def test_c3_model_converged_solves_net_gradients_within_1e_3(dev):
    """The same comparison where BOTH runs have converged (cg_tolerance 1e-3, n = 60 000 -- the size of the dense-truth tests of this model): the
    NET hyper-parameter gradients, the quantities a user sees, agree to 1e-3 -- the bar ``north_star`` states.  (At n = 500 000 and the looser
    tolerance 0.05 below, the two runs are different iterates of one sequence and the net outputscale derivative -- a 3-5-fold cancellation of its
    log-determinant and data-fit parts -- is only within 5e-3: asserted there per part, advisor finding of round 5.)"""
    log, a, b, ysol_err = _run_both_precisions(dev, 60_000, 1e-3, 1000)
    os.makedirs("gpurun_out", exist_ok=True)
    log["grad_net_relative_difference"] = {}
    assert abs(a["inv_quad"] - b["inv_quad"]) < 1e-4 * abs(b["inv_quad"]), log
    assert ysol_err < 1e-3, log
    assert abs(a["logdet"] - b["logdet"]) < 1e-3 * abs(b["logdet"]), log
    for k, name in enumerate(("lengthscale", "outputscale", "noise")):
        ga, gb = a["grad_lengthscale_outputscale_noise"][k], b["grad_lengthscale_outputscale_noise"][k]
        log["grad_net_relative_difference"][name] = abs(ga - gb) / abs(gb)
    with open("gpurun_out/c3_model_converged_vs_float64.json", "w") as f:
        json.dump(log, f, indent=1)
    for name, v in log["grad_net_relative_difference"].items():
        assert v < 1e-3, (name, log)


def test_c3_mll_ingredients_fused_float32_vs_float64_bbmm(dev):
    log, a, b, ysol_err = _run_both_precisions(dev, 500_000, 0.05, 400)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/c3_at_size_vs_float64.json", "w") as f:
        json.dump(log, f, indent=1)

    # the stopping iteration: the mean residual creeps towards the tolerance on a plateau, so rounding moves the crossing by several iterations
    # (measured at tolerance 0.05: 174 float32 against 157 float64 iterations with every ingredient below within 1.5e-4) -- bounded loosely, recorded
    assert abs(a["iterations"] - b["iterations"]) <= 0.2 * b["iterations"], log
    assert abs(a["inv_quad"] - b["inv_quad"]) < 1e-3 * abs(b["inv_quad"]), log
    # (at a LOOSE tolerance the two runs are different iterates of the same sequence -- 0.2: 113 against 98 iterations, inv_quad 2.7e-3 apart,
    # profiles/r05_s4_c3_at_size_vs_float64_tol0.2.json -- so the comparison is made where both have converged to the level it asserts)
    # the y column: both runs stop at a mean relative residual of 5 %, 12 % of their iterations apart -- the two iterates differ by 0.95e-3 .. 1.8e-3 (rel. L2, three
    # runs: profiles/r05_s3_*, r05_s7_*, r05_s8_c3_at_size_*); bounded at 5e-3, the quadratic form y^T K^-1 y above (4e-5) is the converged quantity
    assert ysol_err < 5e-3, log
    assert abs(a["logdet_precond_part"] - b["logdet_precond_part"]) < 1e-6 * abs(b["logdet_precond_part"]), log
    # the quadrature of the fixed probes: relative to the log-determinant it contributes to (its own scale: n times a per-datum O(1) quantity)
    assert abs(a["logdet_slq_part"] - b["logdet_slq_part"]) < 1e-3 * abs(b["logdet"]), log
    assert abs(a["logdet"] - b["logdet"]) < 1e-3 * abs(b["logdet"]), log
    # Gradients: every derivative is the sum of a log-determinant (probe) part and a data-fit (y) part of opposite sign that cancel 3-5 fold, and at
    # cg_tolerance 0.05 the float32 and float64 runs stop 12 % of their iterations apart (chaotic crossing of a plateau).  Criterion (the one of
    # tests/test_gpu_grad_at_size.py: every part of a real backward within 1e-3 of its scale): |difference| < 1e-3 (|log-det part| + |data-fit part|).
    # Measured on the NET values in the first run of this form (profiles/r05_s7_c3_at_size_vs_float64.json): lengthscale 5.8e-4, outputscale 1.3e-3,
    # noise 1.5e-4 -- recorded, the net outputscale derivative is NOT within 1e-3 at this tolerance.
    log["grad_net_relative_difference"] = {}
    for k, name in enumerate(("lengthscale", "outputscale", "noise")):
        ga, gb = a["grad_lengthscale_outputscale_noise"][k], b["grad_lengthscale_outputscale_noise"][k]
        scale = abs(b["grad_logdet_part"][k]) + abs(b["grad_data_fit_part"][k])
        log["grad_net_relative_difference"][name] = abs(ga - gb) / abs(gb)
        assert abs(ga - gb) < 1e-3 * scale, (name, ga, gb, scale, log)
        assert abs(ga - gb) < 5e-3 * abs(gb), (name, ga, gb, log)          # and the net value itself, loosely
    with open("gpurun_out/c3_at_size_vs_float64.json", "w") as f:
        json.dump(log, f, indent=1)
