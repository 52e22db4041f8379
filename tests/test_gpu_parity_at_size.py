"""GPU parity AT THE BENCHMARKED CONFIGURATIONS (BASELINE.json configs C2-C5 and the metric shape).

The small-shape sweeps of test_gpu_kv.py compare every kernel variant with the oracle; here the very kernel
instantiations the benchmark and the scale checks run (n = 1e5 .. 1e6, 33 / 65 columns, Matern-5/2 d = 10, the
Kronecker MVM of C5) are compared with the float64 oracle on a >= 1000-row sample of their output -- rows from the
first and the last row block plus random ones -- at rel 2e-5 of max |K V|.  The oracle rows follow the reference's
default dense formulas (oracle.kernels.kernel_matmul_rows: kernels/kernel.py:26-60, functions/rbf_covariance.py:14-19,
functions/matern_covariance.py:18-50).

C3 is additionally run END TO END: ExactGP MLL solve with the rank-100 pivoted-Cholesky preconditioner until mBCG
reports ``tolerance_reached``; the TRUE residual |K_hat x - b| / |b| is recomputed with one more fused product.
"""
import math

import pytest
import torch

from oracle import kernels as OK
from tests.util import rel_err

pytestmark = pytest.mark.gpu

CONFIGS = {
    # name: kind, n, d, lengthscale, columns (probes + y)
    "c2": ("rbf", 100_000, 3, 0.25, 65),
    "metric": ("rbf", 500_000, 3, 0.25, 65),
    "metric_default_t": ("rbf", 500_000, 3, 0.25, 11),   # reference default num_trace_samples = 10 (+ y)
    "c3": ("matern52", 500_000, 10, 0.8, 65),
    "c4_share": ("rbf", 1_000_000, 3, 0.25, 33),
    # outside the cloud-centred Gram bound (max |x / l|^2 ~ 200): block-centred expansion on the Hilbert-sorted rows (tests/test_gpu_recenter.py);
    # the reference's Gram-trick distance has no scale limit (kernels/kernel.py:26-49)
    "metric_l0.05": ("rbf", 500_000, 3, 0.05, 65),
    "normal_inputs_l0.3": ("rbf", 500_000, 3, 0.3, 65, "normal"),
}


def synth(n, d, seed=0):
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(n, d, generator=g, dtype=torch.float32)
    y = torch.sin(2 * math.pi * X[:, 0]) + torch.cos(math.pi * X.sum(-1)) + 0.1 * torch.randn(n, generator=g)
    return X, y


def sample_rows(n, k=1024, seed=7):
    g = torch.Generator().manual_seed(seed)
    q = k // 4
    mid = torch.randint(q, n - q, (k - 2 * q,), generator=g)
    return torch.cat([torch.arange(q), mid, torch.arange(n - q, n)]).unique()


@pytest.mark.parametrize("split", [False, True], ids=["f32mfma", "split"])
@pytest.mark.parametrize("name", list(CONFIGS))
def test_kv_rows_vs_oracle_at_config(name, split, dev, monkeypatch):
    """Both contraction paths at every benchmarked configuration: the float32-MFMA kernels (kv_gram.hpp) and the split-operand
    kernels on the f16 matrix pipe (kv_gramh.hpp, the default) -- same 2e-5 bound against the float64 oracle."""
    from gpytorch_amd import backend as B

    monkeypatch.setattr(B, "SPLIT_CONTRACTION", split)

    kind, n, d, ls, t = CONFIGS[name][:5]
    X, _ = synth(n, d)
    if len(CONFIGS[name]) > 5:
        X = torch.randn(n, d, generator=torch.Generator().manual_seed(0), dtype=torch.float32)   # standardised inputs
    V = torch.randn(n, t, generator=torch.Generator().manual_seed(1), dtype=torch.float32)
    Xd = X.to(dev)
    xp = B.prep_points(kind, Xd, torch.tensor([ls]), Xd.mean(0))
    if kind != "matern12":
        # the Gram-form instantiation (what bench.py / scale_check.py launch), with / without the split contraction
        assert B.kv_flags(xp, xp, t) == (B.KV_GRAM | (B.KV_SPLIT if split else 0))
        assert B.gram_mode(xp, xp) == (2 if name in ("metric_l0.05", "normal_inputs_l0.3") else 1)
    out_t = B.kv(xp, xp, B.to_probe_major(V.to(dev)))
    rows = sample_rows(n)
    got = out_t[:, rows.to(dev)].t().double().cpu()
    ref = OK.kernel_matmul_rows(kind, X.double(), rows, ls, 1.0, V.double())
    err = rel_err(got, ref)
    assert err < 2e-5, (name, err)
    # a launch with nothing to contract must not touch the sample either: columns are independent
    assert got.shape == (rows.numel(), t)


def test_c5_kronecker_mvm_rows_vs_oracle(dev):
    """C5: (K_XX (x) K_TT) V at n = 200 000, d = 6, T = 4, 17 columns (68 columns in the fused launch)."""
    from gpytorch_amd import backend as B
    from gpytorch_amd.multitask import kron_matvec

    n, d, T, t, ls = 200_000, 6, 4, 17, 0.5
    X, _ = synth(n, d)
    g = torch.Generator().manual_seed(2)
    Bf = torch.randn(T, 1, generator=g, dtype=torch.float64)
    ktt = Bf @ Bf.t() + 0.5 * torch.eye(T, dtype=torch.float64)
    V = torch.randn(n * T, t, generator=g, dtype=torch.float32)
    Xd = X.to(dev)
    xp = B.prep_points("rbf", Xd, torch.tensor([ls]), Xd.mean(0))
    out_t = kron_matvec(xp, xp, ktt.float().to(dev), B.to_probe_major(V.to(dev)))   # [t, n*T] interleaved
    rows = sample_rows(n, 512)
    # oracle: rows (i, :) of K_XX V_mat K_TT^T with V_mat = V reshaped (n, T*t)
    kx = OK.kernel_matmul_rows("rbf", X.double(), rows, ls, 1.0, V.double().reshape(n, T * t))      # [r, T*t]
    ref = torch.einsum("ab,rbc->rac", ktt, kx.reshape(-1, T, t))                                    # [r, T, t]
    idx = (rows.unsqueeze(-1) * T + torch.arange(T)).reshape(-1).to(dev)
    got = out_t[:, idx].t().double().cpu().reshape(-1, T, t)
    assert rel_err(got, ref) < 2e-5


def _true_residual(xp, sc, s2, sol_t, rhs_t):
    from gpytorch_amd import backend as B

    n = xp.n
    res = B.kv(xp, xp, sol_t, scale=sc, dscale=s2, vd=sol_t)[:, :n] - rhs_t[:, :n]
    return res.norm(dim=-1) / rhs_t[:, :n].norm(dim=-1)


def test_c3_end_to_end_preconditioned_mll(dev):
    """BASELINE C3: Matern-5/2, n = 500 000, d = 10, rank-100 pivoted-Cholesky preconditioner, 16 probes + y.
    The reference's training tolerance (cg_tolerance = 1) is reached, the solver's reported residuals are the true
    ones, the preconditioner does not change the solution of the y column and reduces the iteration counts."""
    import json
    import os
    import time

    from gpytorch_amd import backend as B
    from gpytorch_amd.bbmm import LOG_2PI, build_preconditioner, inv_quad_logdet_forward

    kind, n, d, ls = "matern52", 500_000, 10, 0.8
    X, y = synth(n, d)
    Xd, yd = X.to(dev), y.to(dev)
    xp = B.prep_points(kind, Xd, torch.tensor([ls]), Xd.mean(0))
    sc, s2 = torch.tensor([1.0], device=dev), torch.tensor([0.1], device=dev)
    rhs_t = B.to_probe_major(yd.unsqueeze(-1))
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    pre = build_preconditioner(xp, sc, s2, rank=100, min_size=0)
    torch.cuda.synchronize(dev)
    pre_seconds = time.perf_counter() - t0
    assert pre is not None and pre.q1t.shape[0] == 100
    log = {"preconditioner_build_seconds": pre_seconds}
    sols = {}
    for tag, p in (("precond100", pre), ("noprecond", None)):
        gen = torch.Generator(device=dev).manual_seed(11)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        res = inv_quad_logdet_forward(xp, sc, s2, rhs_t, num_probes=16, precond=p, generator=gen, tolerance=1.0, max_iter=600)
        torch.cuda.synchronize(dev)
        seconds = time.perf_counter() - t0
        assert res.info.tolerance_reached, (tag, res.info.iterations, float(res.info.residual_norms.mean()))
        full = torch.cat([res.zt, rhs_t], 0)
        rel = _true_residual(xp, sc, s2, res.solves_t, full)
        rep = res.info.residual_norms
        # reported (recurrence) residuals == true residuals: float32 drift over <= 600 iterations stays below 5 %
        assert float((rel - rep).abs().max()) < 0.05 * max(1.0, float(rep.max())), (tag, rel.tolist(), rep.tolist())
        assert float(rel.mean()) < 1.0
        mll = -0.5 * (float(res.inv_quad.sum()) + float(res.logdet) + n * LOG_2PI) / n
        assert math.isfinite(mll)
        sols[tag] = res
        log[tag] = dict(iterations=res.info.iterations, mean_true_rel_residual=float(rel.mean()), mll=mll,
                        inv_quad=float(res.inv_quad.sum()), logdet=float(res.logdet), mll_evaluation_seconds=seconds)
    # tighter solve of the y column alone (eval tolerance): with and without the preconditioner -> same solution
    from gpytorch_amd.linear_cg import linear_cg

    ysol = {}
    for tag, p in (("precond100", pre), ("noprecond", None)):
        s_t, info = linear_cg(xp, sc, s2, rhs_t, tolerance=0.01, max_iter=2000, preconditioner=p)
        assert info.tolerance_reached, (tag, info.iterations)
        rel = _true_residual(xp, sc, s2, s_t, rhs_t)
        assert float(rel.max()) < 0.012, (tag, float(rel.max()))
        ysol[tag] = s_t
        log[tag]["y_solve_iterations_tol0.01"] = info.iterations
    # both are 1 %-residual solutions of the same system: they agree to a few kappa-free percent in the K_hat norm;
    # compare the quadratic forms y^T K^-1 y they imply
    q = {k_: float((v[0, :n].double() * yd.double()).sum()) for k_, v in ysol.items()}
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/c3_end_to_end.json", "w") as f:
        json.dump(log, f, indent=1)
    assert abs(q["precond100"] - q["noprecond"]) < 2e-3 * abs(q["noprecond"]), q
    # the preconditioner pays (float64 apply, linear_cg.Preconditioner.apply_): measured 44 vs 184 iterations for the 17-column MLL
    # solve, 244 vs 449 for the y column at tolerance 0.01 (profiles/r02_s9_c3_end_to_end.json); with the float32 apply of round 1
    # it did not (465 vs 449)
    assert log["precond100"]["y_solve_iterations_tol0.01"] <= log["noprecond"]["y_solve_iterations_tol0.01"]
    assert log["precond100"]["iterations"] <= log["noprecond"]["iterations"]


def test_c3_miniature_vs_dense_cholesky(dev):
    """The same pipeline (Matern-5/2, d = 10, rank-100 preconditioner, probes from N(0, P)) at n = 3000 against the
    dense float64 answer: inv_quad 1e-3, log-det within the SLQ sampling error (64 probes: 2 %)."""
    from gpytorch_amd import backend as B
    from gpytorch_amd.bbmm import build_preconditioner, inv_quad_logdet_forward
    from oracle import exact_gp as OG

    kind, n, d, ls = "matern52", 3000, 10, 0.8
    X, y = synth(n, d)
    Xd, yd = X.to(dev), y.to(dev)
    xp = B.prep_points(kind, Xd, torch.tensor([ls]), Xd.mean(0))
    sc, s2 = torch.tensor([1.0], device=dev), torch.tensor([0.1], device=dev)
    pre = build_preconditioner(xp, sc, s2, rank=100, min_size=0)
    gen = torch.Generator(device=dev).manual_seed(5)
    res = inv_quad_logdet_forward(xp, sc, s2, B.to_probe_major(yd.unsqueeze(-1)), num_probes=64, precond=pre, generator=gen, tolerance=1e-3)
    sol, ld = OG.dense_solve_logdet(kind, X.double(), y.double().unsqueeze(-1), ls, 1.0, 0.1)
    iq = float((sol.squeeze(-1) * y.double()).sum())
    assert abs(float(res.inv_quad.sum()) - iq) < 1e-3 * abs(iq)
    assert abs(float(res.logdet) - float(ld)) < 0.02 * abs(float(ld))


def test_c4_single_gpu_share_end_to_end(dev):
    """BASELINE C4 as ONE of its 8 ranks sees it: n = 1 000 000, d = 3, RBF, 32 of the 256 probes + the y column (rank 0's share;
    `bench.py --config c4`), no preconditioner, reference-default training tolerance.  The solve reaches the tolerance, the
    recurrence residuals are the true ones (one more fused product), the operator is symmetric on the solution vectors."""
    import json
    import os

    from gpytorch_amd import backend as B
    from gpytorch_amd.bbmm import LOG_2PI, inv_quad_logdet_forward

    n, d, t = 1_000_000, 3, 32
    X, y = synth(n, d)
    Xd, yd = X.to(dev), y.to(dev)
    xp = B.prep_points("rbf", Xd, torch.tensor([0.25]), Xd.mean(0))
    sc, s2 = torch.tensor([1.0], device=dev), torch.tensor([0.1], device=dev)
    rhs_t = B.to_probe_major(yd.unsqueeze(-1))
    gen = torch.Generator(device=dev).manual_seed(1234)
    res = inv_quad_logdet_forward(xp, sc, s2, rhs_t, num_probes=t, precond=None, generator=gen, t_total=256)
    assert res.info.tolerance_reached
    full = torch.cat([res.zt, rhs_t], 0)
    rel = _true_residual(xp, sc, s2, res.solves_t, full)
    rep = res.info.residual_norms
    assert float((rel - rep).abs().max()) < 0.05 * max(1.0, float(rep.max()))
    assert float(rel.mean()) < 1.0
    # u^T (K_hat v) == v^T (K_hat u) on the first two solution vectors
    u, v = res.solves_t[0:1], res.solves_t[1:2]
    ku = B.kv(xp, xp, u, scale=sc, dscale=s2, vd=u)
    kv_ = B.kv(xp, xp, v, scale=sc, dscale=s2, vd=v)
    a, b = float((v.double() * ku.double()).sum()), float((u.double() * kv_.double()).sum())
    # (K is symmetric only to its own accuracy, <= 2e-5 relative; the signed sums cancel heavily, so the bound takes the absolute
    # values inside the product: 2e-5 * |v|^T K_hat |u|)
    kabs = B.kv(xp, xp, u.abs(), scale=sc, dscale=s2, vd=u.abs())
    assert abs(a - b) < 2e-5 * float((v.double().abs() * kabs.double()).sum())
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/c4_share_end_to_end.json", "w") as f:
        json.dump(dict(n=n, columns=t + 1, iterations=res.info.iterations, mean_true_rel_residual=float(rel.mean()),
                       inv_quad=float(res.inv_quad.sum()), logdet_share=float(res.logdet_pinvk),
                       mll_share=-0.5 * (float(res.inv_quad.sum()) + float(res.logdet) + n * LOG_2PI) / n), f, indent=1)


def test_c5_multitask_end_to_end(dev):
    """BASELINE C5 on one GPU: multitask ExactGP, 4 tasks, RBF (x) index kernel, n = 200 000, d = 6 (800 000 rows), batched CG over
    the Kronecker MVM with 16 probes + y.  The solve reaches the tolerance; the TRUE residual of the y solve, recomputed through the
    Kronecker operator, is at the requested tolerance; the operator is symmetric."""
    import json
    import os

    from gpytorch_amd import backend as B
    from gpytorch_amd.multitask import kron_matvec
    from gpytorch_amd.linear_cg import linear_cg

    n, d, T, ls = 200_000, 6, 4, 0.5
    X, _ = synth(n, d)
    g = torch.Generator().manual_seed(2)
    Bf = torch.randn(T, 1, generator=g)
    ktt = (Bf @ Bf.t() + 0.5 * torch.eye(T)).to(dev)
    Y = torch.stack([torch.sin((k + 1.0) * X[:, 0] * 3) + 0.1 * torch.randn(n, generator=g) for k in range(T)], -1).reshape(-1)   # interleaved i*T + tau
    Xd = X.to(dev)
    xp = B.prep_points("rbf", Xd, torch.tensor([ls]), Xd.mean(0))
    N = n * T
    ld = B.round_up(N, 4)
    dv = torch.full((ld,), 0.1, device=dev)
    dv[N:] = 0

    def partials(dt):
        out = kron_matvec(xp, xp, ktt, dt, None)
        return out, 1, out.stride(0)

    t = 16
    rhs = torch.zeros(t + 1, ld, device=dev)
    rhs[:t, :N] = torch.randint(0, 2, (t, N), generator=g).float().to(dev) * 2 - 1
    rhs[t, :N] = Y.to(dev)
    sol, info = linear_cg(None, None, None, rhs, n_tridiag=t, tolerance=0.01, max_iter=1000, kv_partials=partials, dvec=dv, nvec=N)
    assert info.tolerance_reached, info.iterations
    khx = kron_matvec(xp, xp, ktt, sol, None)[:, :N] + 0.1 * sol[:, :N]
    rel = (khx - rhs[:, :N]).norm(dim=-1) / rhs[:, :N].norm(dim=-1)
    assert float(rel.mean()) < 0.012, rel.tolist()
    u, v = sol[0:1], sol[t : t + 1]
    ku = kron_matvec(xp, xp, ktt, u, None)[:, :N].double()
    a = float((v[:, :N].double() * ku).sum())
    b = float((u[:, :N].double() * kron_matvec(xp, xp, ktt, v, None)[:, :N].double()).sum())
    # v^T K u - u^T K v = sum_ij v_i u_j (K_ij - K_ji): the generated K is symmetric only to its own accuracy (<= 2e-5 relative,
    # the squared distances of (i, j) and (j, i) round differently), so the bound is 2e-5 * |v|^T K |u| -- with the absolute
    # values INSIDE the product (the signed sums cancel heavily)
    kabs = kron_matvec(xp, xp, ktt, u.abs(), None)[:, :N].double()
    assert abs(a - b) < 2e-5 * float((v[:, :N].double().abs() * kabs).sum())
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/c5_end_to_end.json", "w") as f:
        json.dump(dict(n=n, tasks=T, rows=N, columns=t + 1, iterations=info.iterations, mean_true_rel_residual=float(rel.mean())), f, indent=1)
