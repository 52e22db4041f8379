#!/usr/bin/env python3
"""Headline benchmark of the BBMM hot path (contract: see task statement / DESIGN.md "Measurement").

One "step" = one ExactGP marginal-log-likelihood evaluation on synthetic data of the configuration
BASELINE.json's `metric` is quoted on -- n = 500 000, d = 3, RBF -- in the single-GPU shape of configs[1]
(64 probe vectors + the y column per GPU, fused K*V HIP kernel, no preconditioner, reference-default
cg_tolerance = 1): probe draw, fused K*V (MFMA) x CG iterations, device-resident CG vector updates, SLQ
log-det, inverse quadratic form.  It fits one GPU (K is never formed: ~1.5 GB of HBM).  `--size 100000` runs
configs[1] itself.  Inputs are resident in HBM before the timed region.

  value      = algorithmic K*V flops of the step (2 n^2 (t+1) x CG iterations, summed over ranks)
               / max-over-ranks wall time                                   [TFLOP/s]
  roofline   = the dominant kernel (kv_mfma) timed live with HIP events on its launch stream:
               2 n^2 (t+1) flop per launch / mean launch duration vs 157.3 TF fp32 MFMA peak
  cpu_baseline = the oracle's matrix-free K*V (the reference's chunked path,
               lazy_evaluated_kernel_tensor.py:245-275, restated in torch) timed on the host cores
               on a bounded row-sample of the same K*V

  parity     = rows of the benchmarked K*V (same kernel instantiation, same n and column count, probe-like V) compared
               with the float64 oracle OUTSIDE the timed region: max |GPU - oracle| / max |oracle|

`--config c3` = BASELINE configs[2] (Matern-5/2, n = 500 000, d = 10, rank-100 pivoted-Cholesky preconditioner; the preconditioner build
is inside the timed step, as in a real MLL evaluation); `--config c5` = BASELINE configs[4] (multitask, 4 tasks, RBF (x) index kernel,
n = 200 000, d = 6: mBCG over the Kronecker MVM -- ONE fused launch with 4 x columns -- 64 probes IN TOTAL split over the ranks).

N > 1: one process per GPU.  Default (`--scaling strong`): the workload BASELINE names -- ONE MLL evaluation, 64 probes in total + y -- split over
the N GPUs on the P x R grid (probe shares x row blocks) `gpytorch_amd.distributed.choose_grid` picks from the measured column ladder of the
fused K*V and the cost of one all-gather of the search directions per iteration: 1 x N for `metric` (every rank keeps all 65 columns on 1/N of
the rows; probe sharding alone would leave 8 + 1 columns per GPU, where kernel generation no longer hides under the contraction: 2x, not 8x).
`--scaling weak` = rounds 1-5's default (64 probes PER GPU, probes only).  `--config c4` = BASELINE configs[3]: n = 1 000 000, 256 probes in
total (2 x 4 on 8 GPUs).  `--grid PxR` overrides the layout.  Data-path collectives: the 2-float stopping-rule all-reduce per CG iteration over the
probe group (stream-ordered RCCL), one scalar SLQ all-reduce, one broadcast of the y solve; with row blocks one all-gather of the search
directions (4 n cols bytes) + two small all-reduces of the inner products per iteration over the row group.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md (v_mfma_f32_32x32x2_f32)
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense f16 MFMA peak (same guide; the sparsity figures are not used)


def synth(n, d, seed=0):
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(n, d, generator=g, dtype=torch.float32)
    y = torch.sin(2 * math.pi * X[:, 0]) + torch.cos(math.pi * X.sum(-1)) + 0.1 * torch.randn(n, generator=g)
    return X, y


def cpu_baseline(n, d, t, ls, budget_s=12.0, kind="rbf"):
    """The reference's matrix-free K*V on the host cores, as SURVEY.md 8(d) prescribes: float32 (the reference computes in the
    dtype of its inputs), row chunks of 4096 (`lazy_evaluated_kernel_tensor.py:245-275`: split x1, build the chunk of K with
    the kernel's own forward -- the mean-centred Gram-trick `sq_dist` of `kernels/kernel.py:26-49` -- multiply, cat), restated
    by the oracle.  Timed on a bounded sample: whole 4096-row chunks of ONE n x n product until ~budget_s seconds."""
    import psutil

    from oracle import kernels as OK

    X, _ = synth(n, d)
    V = torch.randn(n, t, generator=torch.Generator().manual_seed(1))
    chunk = 4096
    # a 4096 x n float32 chunk of K plus the temporaries of sq_dist / exp is ~4 chunk-sized buffers
    if psutil.virtual_memory().available < 6 * chunk * n * 4:
        chunk = max(64, int(psutil.virtual_memory().available // (6 * n * 4)) // 64 * 64)
    rows_done, t0 = 0, time.perf_counter()
    while rows_done < n and (rows_done == 0 or time.perf_counter() - t0 < budget_s):
        xc = X[rows_done : rows_done + chunk]
        if kind == "rbf":
            kc = OK.rbf(xc, X, ls, x1_eq_x2=False)      # functions/rbf_covariance.py:14-19 on kernels/kernel.py:26-49
        else:
            kc = OK.matern(xc, X, ls, OK.KINDS[kind], x1_eq_x2=False)   # functions/matern_covariance.py:18-50
        _ = kc @ V
        rows_done += xc.shape[0]
    dt = time.perf_counter() - t0
    return {
        "value": 2.0 * rows_done * n * t / dt / 1e12,
        "unit": "TFLOP/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "dtype": "f32",
        "sample": f"port of the reference's chunked matrix-free K*V, sampled: rows 0:{rows_done} of one n={n} product with t={t} in "
                  f"{chunk}-row chunks (float32, Gram-trick sq_dist), {dt:.1f} s on {torch.get_num_threads()} threads",
    }


class ClockSampler:
    """Samples the shader clock (rocm-smi --showclocks, sclk) twice a second in a background thread while a block of steps runs: the f16
    matrix pipe at full load runs well below the 2.4 GHz the fp32 kernel sustains, and the split kernel's time follows that clock."""

    def __init__(self):
        import threading

        self.samples, self._stop = [], threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import re
        import subprocess

        while not self._stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
                m = re.search(r"GPU\[0\]\s*:\s*sclk clock level:\s*\d+:\s*\((\d+)Mhz\)", out)
                if m:
                    self.samples.append(int(m.group(1)))
            except Exception:
                pass
            self._stop.wait(0.5)

    def __enter__(self):
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thread.join(timeout=10)

    def summary(self):
        if not self.samples:
            return None
        v = sorted(self.samples)
        return {"sclk_mhz_median": v[len(v) // 2], "sclk_mhz_min": v[0], "sclk_mhz_max": v[-1], "samples": len(v), "source": "rocm-smi --showclocks"}


def parity_block(xp, Xcpu, n, d, t, ls, dev, nrows=2048, kind="rbf"):
    """Rows of the benchmarked product vs the float64 oracle (reference formulas: oracle.kernels.kernel_matmul_rows)."""
    from gpytorch_amd import backend as B
    from oracle import kernels as OK

    g = torch.Generator().manual_seed(99)
    V = torch.randn(n, t, generator=g)
    V = V / V.norm(dim=0, keepdim=True)                      # probe-like: unit columns, as the CG right-hand sides
    out_t = B.kv(xp, xp, B.to_probe_major(V.to(dev)))
    q = nrows // 4
    rows = torch.cat([torch.arange(q), torch.randint(q, n - q, (nrows - 2 * q,), generator=g), torch.arange(n - q, n)]).unique()
    got = out_t[:, rows.to(dev)].t().double().cpu()
    ref = OK.kernel_matmul_rows(kind, Xcpu.double(), rows, ls, 1.0, V.double())
    return {
        "kv_rel_err": float((got - ref).abs().max() / ref.abs().max()),
        "rows": int(rows.numel()),
        "columns": t,
        "vs": "oracle fp64 (reference dense formulas, kernels/kernel.py:26-49 + functions/" + ("rbf_covariance.py:14-19)" if kind == "rbf" else "matern_covariance.py:18-50)"),
        "tolerance": 2e-5,
    }


def api_level_extras(Xd, yd, ls, t, dev, n_test=10_000):
    """The wall-clock half of BASELINE's metric ("ExactGP MLL+posterior wall-clock"), untimed by the headline: the same workload through the
    gpytorch-shaped API on the library defaults (split contraction).

    * ``mll_by_preconditioner_rank``: ExactMarginalLogLikelihood forward + backward (fused bilinear-derivative kernel) per
      ``settings.max_preconditioner_size`` in {0, 15 = the reference default, 100 = BASELINE config 3's, "auto"}, each with its CG iteration
      count and the deviation of y^T K^-1 y / log|K| from a converged evaluation (rank-100 preconditioner, cg_tolerance 1e-3, same probe count).
    * ``posterior``: cold predictive posterior on ``n_test`` points (both prediction caches missing: mean-cache mBCG + LOVE cache + the K_*X
      products) at the reference defaults (rank-15 preconditioner, eval_cg_tolerance 0.01, LOVE rank 100), with the rank-100 preconditioner,
      and at the settings that meet BASELINE's tolerance (preconditioner "auto" = rank 256 at this size, eval_cg_tolerance 1e-4, LOVE rank 400 -> 12
      block-Lanczos products of 32 columns = rank 384), each with its mean error against a converged float64-refined solve and its variance error in units of the noise
      against the exact-variance path on 64 of the test points (the reference's criterion: < 0.05, test_simple_gp_regression.py:436-442)."""
    import gpytorch_amd as g
    from gpytorch_amd import linear_cg as LCG

    class GPModel(g.models.ExactGP):
        def __init__(self, x, y, lik):
            super().__init__(x, y, lik)
            self.mean_module = g.means.ConstantMean()
            self.covar_module = g.kernels.ScaleKernel(g.kernels.RBFKernel())

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    lik = g.likelihoods.GaussianLikelihood().to(dev)
    m = GPModel(Xd, yd, lik).to(dev)
    m.covar_module.base_kernel.lengthscale = ls
    m.covar_module.outputscale = 1.0
    s2 = 0.1
    lik.noise = s2
    mll = g.ExactMarginalLogLikelihood(lik, m)
    S = g.settings
    n = Xd.shape[0]
    res = {}

    def sync():
        torch.cuda.synchronize(dev)

    def iql(rank, tol):
        m.train(); lik.train()
        with torch.no_grad(), S.max_cholesky_size(0), S.num_trace_samples(t), S.max_preconditioner_size(rank), S.cg_tolerance(tol), S.max_cg_iterations(4000):
            mvn = lik(m(m.train_inputs[0]))
            iq, ld = mvn.lazy_covariance_matrix.evaluate_kernel().inv_quad_logdet((m.train_targets - mvn.mean).unsqueeze(-1), logdet=True)
            return float(iq), float(ld), LCG.LAST_INFO.iterations

    iq_ref, ld_ref, it_ref = iql(100, 1e-3)
    res["mll_converged_reference"] = {"settings": "rank-100 preconditioner, cg_tolerance 1e-3", "inv_quad": iq_ref, "logdet": ld_ref, "cg_iterations": it_ref}
    rows = []
    for rank in (100, 0, 15, "auto"):    # (rank 100 first: its first pass also warms the allocator)
        m.train(); lik.train()
        rec = {"max_preconditioner_size": rank, "resolved_rank": None}
        with S.max_cholesky_size(0), S.num_trace_samples(t), S.max_preconditioner_size(rank):
            rec["resolved_rank"] = S.max_preconditioner_size.resolve(n)
            fw, bw, its = [], [], []
            for rep in range(1 if rank == 0 else 2):
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
                fw.append((t1 - t0) * 1e3); bw.append((t2 - t1) * 1e3); its.append(LCG.LAST_INFO.iterations)
            rec.update(forward_ms=min(fw), backward_ms=min(bw), forward_plus_backward_ms=min(a + b for a, b in zip(fw, bw)), cg_iterations=its[-1],
                       mll_value=float(-loss.detach()), grad_raw_lengthscale=float(m.covar_module.base_kernel.raw_lengthscale.grad.sum()))
        iq, ld, it = iql(rank, 1.0)
        rec.update(inv_quad_rel_dev=abs(iq - iq_ref) / abs(iq_ref), logdet_rel_dev=abs(ld - ld_ref) / abs(ld_ref))
        rows.append(rec)
    rows.sort(key=lambda r: (r["max_preconditioner_size"] == "auto", r["resolved_rank"]))
    res["mll_by_preconditioner_rank"] = rows
    by = {r["max_preconditioner_size"]: r for r in rows}
    # (keys of earlier rounds, kept: no preconditioner)
    res["mll_forward_ms"], res["mll_backward_ms"] = by[0]["forward_ms"], by[0]["backward_ms"]
    res["mll_value"], res["grad_raw_lengthscale"] = by[0]["mll_value"], by[0]["grad_raw_lengthscale"]
    res["mll_forward_plus_backward_ms_auto_preconditioner"] = by["auto"]["forward_plus_backward_ms"]

    Xs, _ = synth(n_test, Xd.shape[-1], seed=3)
    Xs = Xs.to(dev)
    m.eval(); lik.eval()
    nv = 64     # test points of the exact-variance reference (one 64-column solve)
    with torch.no_grad(), S.max_cholesky_size(0), S.eval_cg_tolerance(1e-6), S.skip_posterior_variances(), S.max_preconditioner_size(100), \
            S.max_cg_iterations(4000), S.rhs_refinement():
        mean_ref = m(Xs).mean.double()
    m.train(); m.eval()
    with torch.no_grad(), S.max_cholesky_size(0), S.eval_cg_tolerance(1e-4), S.fast_pred_var(False), S.max_preconditioner_size(100), S.max_cg_iterations(4000):
        var_ref = lik(m(Xs[:nv])).variance.double()
    post = []
    cases = (("reference defaults", 15, 1e-2, 100, "auto"), ("rank-100 preconditioner", 100, 1e-2, 100, "auto"),
             ("meets BASELINE's tolerance", "auto", 1e-4, 400, "auto"))
    for tag, rank, tol, love, blk in cases:
        m.train(); m.eval()   # drops the prediction strategy (caches)
        with torch.no_grad(), S.max_cholesky_size(0), S.fast_pred_var(), S.max_preconditioner_size(rank), S.eval_cg_tolerance(tol), \
                S.max_root_decomposition_size(love), S.lanczos_block_size(blk), S.max_cg_iterations(4000):
            # "cold" = the model's caches dropped, not the operating system's: the first evaluation of a configuration can also page in solver-library
            # code objects from disk (one run on a fresh box read 4.35 s for the third case where five others read 1.0 - 1.2 s) -- two cold
            # evaluations, the faster one reported, both kept
            cold_all = []
            for _ in range(2):
                m.train(); m.eval()
                sync()
                t0 = time.perf_counter()
                pred = lik(m(Xs))
                mu, var = pred.mean, pred.variance
                sync()
                t1 = time.perf_counter()
                cold_all.append((t1 - t0) * 1e3)
            its = LCG.LAST_INFO.iterations if LCG.LAST_INFO is not None else None
            pred = lik(m(Xs))  # caches warm
            mu, var = pred.mean, pred.variance
            sync()
            t2 = time.perf_counter()
            rank_res = S.max_preconditioner_size.resolve(n)
        post.append({"settings": tag, "max_preconditioner_size": rank, "resolved_rank": rank_res, "eval_cg_tolerance": tol, "love_rank": love,
                     "lanczos_block_size": blk, "cold_ms": min(cold_all), "cold_ms_all": cold_all, "warm_ms": (t2 - t1) * 1e3, "mean_cache_cg_iterations": its,
                     "mean_max_err_over_max_abs_mean": float((mu.double() - mean_ref).abs().max() / mean_ref.abs().max()),
                     "var_max_err_over_noise": float((var.double()[:nv] - var_ref).abs().max() / s2)})
    res["posterior"] = post
    res["posterior_references"] = {"mean": "eval_cg_tolerance 1e-6 + one float64 refinement step (settings.rhs_refinement), rank-100 preconditioner",
                                   "variance": f"exact-variance path (fast_pred_var off) on the first {nv} test points, eval_cg_tolerance 1e-4"}
    # (keys of earlier rounds, kept)
    res["posterior_cold_ms"], res["posterior_warm_ms"] = post[0]["cold_ms"], post[0]["warm_ms"]
    res["posterior_mean_cache_cg_iterations"] = post[0]["mean_cache_cg_iterations"]
    res["posterior_cold_precond100_ms"], res["posterior_warm_precond100_ms"] = post[1]["cold_ms"], post[1]["warm_ms"]
    res["posterior_mean_cache_cg_iterations_precond100"] = post[1]["mean_cache_cg_iterations"]
    res["posterior_cold_accurate_ms"] = post[2]["cold_ms"]
    res["posterior_test_points"] = n_test
    res["posterior_mean_abs_max"] = float(mu.abs().max())
    res["posterior_var_min_max"] = [float(var.min()), float(var.max())]
    return res


def batch_small_extras(dev, b=64, n=200, d=3, reps=3):
    """A batch of SMALL exact GPs (members below max_cholesky_size): one MLL evaluation + backward through the stacked path
    (gpytorch_amd/batched.py: one dense-generation launch, batched Cholesky, one derivative launch) and through the launch plan over
    members.  Same measurement as scripts/batch_small_timing.py."""
    import gpytorch_amd as g

    bs = torch.Size([b])
    gen = torch.Generator().manual_seed(0)
    X = torch.rand(b, n, d, generator=gen).to(dev)
    Y = (torch.sin(3 * X.sum(-1).cpu()) + 0.1 * torch.randn(b, n, generator=gen)).to(dev)

    class M(g.models.ExactGP):
        def __init__(self, x, y, lik):
            super().__init__(x, y, lik)
            self.mean_module = g.means.ConstantMean(batch_shape=bs)
            self.covar_module = g.kernels.ScaleKernel(g.kernels.RBFKernel(batch_shape=bs), batch_shape=bs)

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    lik = g.likelihoods.GaussianLikelihood(batch_shape=bs).to(dev)
    m = M(X, Y, lik).to(dev)
    mll = g.ExactMarginalLogLikelihood(lik, m)
    m.train(); lik.train()
    res = {"members": b, "points_per_member": n, "d": d}
    for stacked in (True, False):
        with g.settings.batched_small_members(stacked):
            best = None
            for _ in range(reps):
                m.zero_grad()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                val = mll(m(X), Y).sum()
                val.backward()
                torch.cuda.synchronize(dev)
                dt = (time.perf_counter() - t0) * 1e3
                best = dt if best is None else min(best, dt)
            res["stacked_ms" if stacked else "member_loop_ms"] = best
            res["mll_sum_stacked" if stacked else "mll_sum_member_loop"] = float(val.detach())
    return res


def self_launch(n_ranks: int) -> int:
    """Re-execute this script under ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`` with the same
    arguments; stdout / stderr pass through (rank 0 prints the JSON line); returns the launcher's exit code."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n_ranks)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def make_grid_groups(world: int, rank: int, P: int, R: int):
    """--grid PxR: P probe shares x R row blocks, rank = p * R + r.  Returns (probe group of this rank = the P ranks with its r -- stopping rule,
    SLQ sums --, row group = the R ranks with its p -- all-gather of the search directions, inner products); ``None`` for a dimension of
    extent 1 (``gpytorch_amd.distributed.grid_groups``: collective, every rank creates every group in the same order)."""
    from gpytorch_amd.distributed import grid_groups

    assert P * R == world, f"--grid {P}x{R} needs {P * R} ranks, have {world}"
    return grid_groups(P, R)


# nominal (n, probes in total) of every configuration: the layout of a multi-GPU run is chosen for THESE, so that a --size smoke run exercises
# the layout of the real one
NOMINAL = {"metric": (500_000, 64), "c2": (100_000, 64), "c3": (500_000, 64), "c4": (1_000_000, 256), "c5": (200_000, 64)}


def scaling_of(args) -> str:
    return "strong" if (args.config in ("c4", "c5") or args.scaling == "strong") else "weak"


def pick_grid(args, world: int):
    """(P probe shares, R row blocks) of this run: --grid, else probes only for c5 (structured operator: rows cannot be sharded) and for weak
    scaling, else the cost model's choice for the configuration's nominal size and the timed contraction."""
    if args.grid:
        P, R = (int(v) for v in args.grid.lower().split("x"))
        assert P * R == world, f"--grid {args.grid} needs {P * R} ranks, have {world}"
        return P, R
    if world == 1 or args.config == "c5" or scaling_of(args) == "weak":
        return world, 1
    from gpytorch_amd.distributed import choose_grid

    n_nom, t_nom = NOMINAL[args.config]
    return choose_grid(world, n_nom, args.probes if args.probes is not None else t_nom, args.contraction)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=["metric", "c2", "c3", "c4", "c5", "road3d", "protein"], default="metric",
                    help="metric: n=500k, 64 probes + y (the configuration BASELINE.json's metric is quoted on; N > 1: see --scaling); "
                         "c2: configs[1] (n=100k); c3: configs[2] (Matern-5/2, n=500k, d=10, rank-100 preconditioner); "
                         "c4: configs[3], n=1e6 with 256 probes IN TOTAL split over the ranks (strong scaling); "
                         "c5: configs[4], 4-task Kronecker multitask GP, n=200k, d=6, 64 probes IN TOTAL split over the ranks")
    ap.add_argument("--size", type=int, default=None, help="override the number of training points n (not --n: torch.distributed.run's parser treats that as an ambiguous prefix)")
    ap.add_argument("--dims", type=int, default=None)
    ap.add_argument("--probes", type=int, default=None, help="override: probes per GPU (metric / c2) or in total (c4)")
    ap.add_argument("--contraction", choices=["f32", "split"], default="f32",
                    help="K*V contraction of the TIMED steps: f32 = v_mfma_f32_32x32x2_f32 (the metric's fp32-MFMA roofline; default), "
                         "split = hi/lo-split operands on the f16 matrix pipe (the library default, settings.split_contraction). "
                         "The other one is measured once, untimed, and reported beside it (block 'split_contraction').")
    ap.add_argument("--skip-split", action="store_true", help="skip the steps on the other contraction path")
    ap.add_argument("--other-steps", type=int, default=3, help="steps timed on the other contraction path (reported beside the headline, with spread and the sustained clock)")
    ap.add_argument("--skip-parity", action="store_true")
    ap.add_argument("--skip-cpu-baseline", action="store_true")
    ap.add_argument("--skip-extras", action="store_true", help="skip the untimed API-level MLL fwd+bwd / posterior timings")
    ap.add_argument("--far-cutoff", type=float, default=None, help="--config road3d: run under settings.far_pair_cutoff(eps) (opt-in far-pair tile culling; default: every pair)")
    ap.add_argument("--grid", default=None, help="PxR (N = P * R ranks): two-dimensional split -- P probe shares x R row blocks "
                                                 "(bbmm.inv_quad_logdet_forward(group, row_group)).  Default for metric / c2 / c3 / c4: chosen by the cost model of "
                                                 "gpytorch_amd.distributed.choose_grid at the configuration's nominal size (metric on 8 GPUs: 1x8); c5 and "
                                                 "--scaling weak: probes only (P = N, R = 1)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1, metric / c2 / c3: strong (default) = the workload BASELINE names -- ONE MLL evaluation with 64 probes in total -- split "
                         "over the N GPUs; weak = 64 probes PER GPU (the default of rounds 1-5).  c4 / c5 are strong by definition")
    args = ap.parse_args()

    if args.config in ("road3d", "protein"):
        # the reference's published workloads end to end (training loop + prediction through the gpytorch-shaped API), SURVEY.md 8(f)1:
        # scripts/reference_workloads.py; --size / --steps shrink them for smoke runs
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import reference_workloads

        return reference_workloads.main(args.config, gpus=args.gpus, size=args.size, steps=args.steps if "--steps" in sys.argv else None,
                                        far_eps=args.far_cutoff)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher (the shape of the N = 1 command): start the N ranks ourselves, exactly as the driver's
        # multi-GPU line does -- one process per GPU under torch.distributed.run on the loopback address; rank 0 prints the one JSON line
        return self_launch(args.gpus)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook (one-GPU boxes): GPAMD_BENCH_BACKEND=gloo GPAMD_BENCH_SHARE_DEVICE=1 runs N ranks on cuda:0 with the
    # stopping-rule / SLQ all-reduces carried by gloo -- same code path as RCCL apart from the transport
    backend = os.environ.get("GPAMD_BENCH_BACKEND", "nccl")
    if os.environ.get("GPAMD_BENCH_SHARE_DEVICE") == "1":
        local_rank = 0
    if os.environ.get("GPAMD_BENCH_LAUNCH_ONLY") == "1":
        # launcher test hook (tests/test_bench_launch_cpu.py, no GPU needed): prove that N ranks came up and can talk, then stop
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
            tt = torch.tensor([float(rank + 1)])
            torch.distributed.all_reduce(tt)
            tot = float(tt.item())
            grid = None
            P_, R_ = pick_grid(args, world)
            if R_ > 1:   # the subgroups of the split, exercised over gloo: sums of (rank + 1) over this rank's probe group and row group
                pg, rg = make_grid_groups(world, rank, P_, R_)
                a, b = torch.tensor([float(rank + 1)]), torch.tensor([float(rank + 1)])
                if pg is not None:
                    torch.distributed.all_reduce(a, group=pg)
                if rg is not None:
                    torch.distributed.all_reduce(b, group=rg)
                allv = [None] * world
                torch.distributed.all_gather_object(allv, (rank, float(a.item()), float(b.item())))
                grid = sorted(allv)
            torch.distributed.destroy_process_group()
        else:
            tot, grid = 1.0, None
        if rank == 0:
            rec = {"launched": world, "rank_sum": tot, "n_gpus": args.gpus, "grid": list(pick_grid(args, world)), "scaling": scaling_of(args)}
            if grid is not None:
                rec["grid_sums"] = grid
            print(json.dumps(rec), flush=True)
        return 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            torch.distributed.init_process_group(backend, rank=rank, world_size=world)
        group = torch.distributed.group.WORLD
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # --grid PxR: P probe groups x R row blocks, rank = p * R + r.  The probe group of a rank = the P ranks with its r (stopping rule, SLQ sums),
    # its row group = the R ranks with its p (all-gather of the search directions, inner products).  new_group is collective: every rank
    # creates every group in the same order.
    P_, R_ = pick_grid(args, world)
    row_group = None
    if R_ > 1:
        group, row_group = make_grid_groups(world, rank, P_, R_)
    p_idx = rank // R_          # index of this rank's probe share

    from gpytorch_amd import backend as B
    from gpytorch_amd import linear_cg as LCG
    from gpytorch_amd import settings as gsettings
    from gpytorch_amd.bbmm import LOG_2PI, build_preconditioner, inv_quad_logdet_forward

    gsettings.split_contraction._set_state(args.contraction == "split")

    strong = scaling_of(args) == "strong"
    n = args.size if args.size is not None else {"metric": 500_000, "c2": 100_000, "c3": 500_000, "c4": 1_000_000, "c5": 200_000}[args.config]
    d = args.dims if args.dims is not None else {"c3": 10, "c5": 6}.get(args.config, 3)
    kind = "matern52" if args.config == "c3" else "rbf"
    precond_rank = 100 if args.config == "c3" else 0
    T = 4 if args.config == "c5" else 1
    if strong:
        from gpytorch_amd.distributed import probe_shard

        t_total = args.probes if args.probes is not None else NOMINAL[args.config][1]
        a_, b_ = probe_shard(t_total, P_, p_idx)
        t = b_ - a_
    else:
        t = args.probes if args.probes is not None else 64
        t_total = t * P_
    ls = {3: 0.25, 10: 0.8, 6: 0.5}.get(d, 0.25)
    X, y = synth(n, d)
    Xd, yd = X.to(dev), y.to(dev)
    lengthscale = torch.tensor([ls], device=dev)
    outputscale = torch.tensor([1.0], device=dev)
    noise = torch.tensor([0.1], device=dev)
    gen = torch.Generator(device=dev).manual_seed(1234 + p_idx)   # one probe stream per probe share (row blocks of a share draw the same)
    rhs_t = B.to_probe_major(yd.unsqueeze(-1))
    shift = Xd.mean(dim=0)
    nvec = n * T

    if args.config == "c5":
        # task matrix B = randn(4, 1, seed 2), v = 0.5 -> K_TT = B B^T + 0.5 I; targets: four phase-shifted copies of the
        # synthetic function in the interleaved layout (row = i T + tau, multitask_multivariate_normal.py:66-70); task noise 0.1
        from gpytorch_amd.multitask import kron_matvec

        Bf = torch.randn(T, 1, generator=torch.Generator().manual_seed(2))
        ktt = (Bf @ Bf.t() + 0.5 * torch.eye(T)).to(dev)
        Ymat = torch.stack([yd * math.cos(0.4 * k_) + torch.roll(yd, k_) * math.sin(0.4 * k_) for k_ in range(T)], -1)   # [n, T]
        rhs_t = B.to_probe_major(Ymat.reshape(-1, 1))
        dv = torch.zeros(B.round_up(nvec, 4), device=dev)
        dv[:nvec] = 0.1

    def step():
        xp = B.prep_points(kind, Xd, lengthscale, shift)  # centred, as the kernels' forward does
        if args.config == "c5":
            def partials(dt):
                out = kron_matvec(xp, xp, ktt, dt, outputscale)
                return out, 1, out.stride(0)

            res = inv_quad_logdet_forward(None, None, None, rhs_t, num_probes=t, precond=None, generator=gen, group=group, t_total=t_total,
                                          dvec=dv, kv_partials=partials, nvec=nvec)
        else:
            pre = build_preconditioner(xp, outputscale, noise, rank=precond_rank, min_size=0) if precond_rank else None
            res = inv_quad_logdet_forward(
                xp, outputscale, noise, rhs_t, num_probes=t, precond=pre, generator=gen, group=group, t_total=t_total, row_group=row_group
            )
        mll = -0.5 * (res.inv_quad.sum() + res.logdet + nvec * LOG_2PI) / nvec
        return mll, res.info.iterations

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    LCG.KV_EVENT_LOG = []
    barrier()
    t0 = time.perf_counter()
    iters_total = 0
    iters_all = []
    mll = None
    gen_state = None
    for _ in range(args.steps):
        gen_state = gen.get_state()   # (the untimed step on the other contraction path re-uses the last step's probe draw)
        mll, it = step()
        iters_total += it
        iters_all.append(it)          # (the stopping iteration varies with the probe draw: 93 .. 109 on the metric configuration)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        et = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        torch.distributed.all_reduce(et, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(et.item())

    # dominant-kernel timing from the HIP events recorded on the launch stream inside the timed region
    durs = sorted(e0.elapsed_time(e1) for (e0, e1, _, _, _) in LCG.KV_EVENT_LOG)
    LCG.KV_EVENT_LOG = None
    med = durs[len(durs) // 2]
    live = [x for x in durs if x > 0.2 * med]  # launches issued after convergence are device-side no-ops
    kv_ms = sum(live) / len(live)
    cols = t + (1 if p_idx == 0 else 0)      # the y column is solved by the first probe share only
    rows_loc = n if R_ == 1 else min(n, B.round_up((n + R_ - 1) // R_, 4))   # row block of this rank (distributed.RowShard)
    flop_per_launch = 2.0 * rows_loc * n * cols * T   # c5: the fused launch of the Kronecker MVM carries T x columns (+ O(n T t) glue, inside the events)
    achieved = flop_per_launch / (kv_ms * 1e-3) / 1e12

    # HBM traffic of the dominant kernel: PMC counters cannot be collected from inside this process; the
    # figure comes from the committed rocprofv3 --pmc passes of the same kernel at the same shape
    # (scripts/gpu_session.sh -> scripts/collect_profiles.py -> profiles/kv_pmc_current.json), else null.
    traffic = None
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "kv_pmc_current.json")))
        if prof.get("shape") == [n, d, cols]:
            traffic = prof.get("hbm_bytes_per_launch")
    except Exception:
        traffic = None

    split_traffic = None   # the split kernel's PMC passes (same rules as above): profiles/kv_pmc_split_current.json
    try:
        sp = json.load(open(os.path.join(ROOT, "profiles", "kv_pmc_split_current.json"))).get("_derived", {})
        if [n, d, cols] == [500_000, 3, 65]:
            split_traffic = sp.get("hbm_bytes_per_launch")
    except Exception:
        split_traffic = None

    # whole-job algorithmic flops of the timed region: every rank's columns (t_total probes + one y) x CG iterations
    flops_job = 2.0 * n * n * (t_total + 1) * T * iters_total
    value = flops_job / elapsed / 1e12
    parity = None
    if rank == 0 and not args.skip_parity:
        parity = parity_block(B.prep_points(kind, Xd, lengthscale, shift), X, n, d, cols * T, ls, dev, kind=kind)

    # the other contraction path: --other-steps steps (default 5) on the same inputs after one warm-up, the first of them on the probe draw of
    # the last timed step; its kernel timed the same way (HIP events on the launch stream), the shader clock sampled alongside
    other = None
    if world == 1 and not args.skip_split:
        split_now = args.contraction != "split"
        with gsettings.split_contraction(split_now):
            gen.set_state(gen_state)
            step()
            LCG.KV_EVENT_LOG = []
            gen.set_state(gen_state)      # same probe vectors as the last timed step
            step_ms, its_o = [], []
            mll_o = it_o = None
            with ClockSampler() as clk:
                for k_ in range(max(1, args.other_steps)):
                    torch.cuda.synchronize(dev)
                    t1 = time.perf_counter()
                    mll_k, it_k = step()
                    torch.cuda.synchronize(dev)
                    step_ms.append((time.perf_counter() - t1) * 1e3)
                    its_o.append(it_k)
                    if k_ == 0:
                        mll_o, it_o = mll_k, it_k
            el_o = step_ms[0] * 1e-3
            d_o = sorted(e0.elapsed_time(e1) for (e0, e1, _, _, _) in LCG.KV_EVENT_LOG)
            LCG.KV_EVENT_LOG = None
            live_o = [x for x in d_o if x > 0.2 * d_o[len(d_o) // 2]]
            ms_o = sum(live_o) / len(live_o)
            other = {
                "contraction": "split (hi/lo f16 operands on v_mfma_f32_32x32x16_f16, f32 accumulate; settings.split_contraction, "
                               "the library default)" if split_now else "f32 (v_mfma_f32_32x32x2_f32)",
                "kernel_ms": ms_o,
                "kv_tflops_f32_equivalent": flop_per_launch / (ms_o * 1e-3) / 1e12,
                "speedup_vs_timed_path": kv_ms / ms_o,
                "ms_per_step": el_o * 1e3,
                "steps_timed": len(step_ms),
                "ms_per_step_all": step_ms,
                "cg_iterations_all": its_o,
                "ms_per_cg_iteration_all": [a_ / max(b_, 1) for a_, b_ in zip(step_ms, its_o)],
                "kernel_ms_min_median_max": [live_o[0], live_o[len(live_o) // 2], live_o[-1]],
                "clock": clk.summary(),
                "cg_iterations": it_o,
                "cg_iterations_timed_path_same_probes": it,
                "mll": float(mll_o),
                "mll_timed_path_same_probes": float(mll),
                "launches_timed": len(live_o),
            }
            if split_now:
                # three f16 MFMAs per f32-equivalent multiply-add: executed flops against the dense f16 peak
                ex = 3.0 * flop_per_launch / (ms_o * 1e-3) / 1e12
                other["roofline"] = {"bound": "mfma", "achieved": ex, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s (f16, executed = 3 x algorithmic)",
                                     "frac": ex / PEAK_F16_MFMA_TFLOPS, "traffic": split_traffic,
                                     "traffic_source": "from profiles/kv_pmc_split_current.json (recorded constant)" if split_traffic is not None else None,
                                     "kernel": "kv_gramh_kernel (csrc/kv_gramh.hpp)"}
            if not args.skip_parity:
                other["parity"] = parity_block(B.prep_points(kind, Xd, lengthscale, shift), X, n, d, cols * T, ls, dev, kind=kind)

    extras = None
    if world == 1 and not args.skip_extras and args.config in ("metric", "c2"):
        gsettings.split_contraction._set_state(None)   # API-level timings on the library defaults (split contraction on)
        extras = api_level_extras(Xd, yd, ls, t, dev)
        extras["kv_contraction"] = "library default: split" if gsettings.split_contraction.on() else "f32"
        try:
            extras["batch_of_small_gps"] = batch_small_extras(dev)
        except Exception as e:  # (reported, never fatal for the headline line)
            extras["batch_of_small_gps"] = {"error": repr(e)[:200]}

    if args.contraction == "f32":
        roofline = {
            "bound": "mfma",
            "achieved": achieved,
            "peak": PEAK_FP32_MFMA_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
            "traffic": traffic,
            "traffic_unit": "bytes per launch (2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes)",
            "traffic_source": "from profiles/kv_pmc_current.json (rocprofv3 --pmc passes of the builder's sessions on the same kernel and shape: a recorded constant, "
                              "not measured in this process)" if traffic is not None else None,
            "kernel": f"kv_gram_kernel<{kind},D={d},CT={(cols * T - 1) // 32 if (cols * T) % 32 == 1 else (cols * T + 31) // 32},EX={1 if (cols * T) % 32 == 1 else 0}> on rank 0 (Gram-form generation "
                      "on split-f16 MFMA + fp32 MFMA contraction; kv_mfma_kernel when max|z|^2 > 32)",
            "kernel_ms": kv_ms,
            "launches_timed": len(live),
            "flop_per_launch": flop_per_launch,
        }
    else:
        # three f16 MFMAs per f32-equivalent multiply-add: EXECUTED flops (3 x algorithmic) against the dense f16 peak
        roofline = {
            "bound": "mfma",
            "achieved": 3.0 * achieved,
            "peak": PEAK_F16_MFMA_TFLOPS,
            "unit": "TFLOP/s (f16 MFMA, executed = 3 x algorithmic)",
            "frac": 3.0 * achieved / PEAK_F16_MFMA_TFLOPS,
            "traffic": split_traffic,
            "traffic_unit": "bytes per launch (2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes)",
            "traffic_source": "from profiles/kv_pmc_split_current.json (recorded constant, not measured in this process)" if split_traffic is not None else None,
            "kernel": f"kv_gramh_kernel<{kind},D={d}> on rank 0 (Gram-form generation and hi/lo-split contraction on v_mfma_f32_32x32x16_f16)",
            "kernel_ms": kv_ms,
            "launches_timed": len(live),
            "flop_per_launch": flop_per_launch,
            "algorithmic_tflops": achieved,
        }
    if rank == 0:
        out = {
            "metric": "exactgp_mll_kv_tflops",
            "value": value,
            "unit": "TFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",   # (N = 1: the same workload either way)
            "vs_baseline": None,
            "dtype": "f32" if args.contraction == "f32" else "f32 emulated on f16 MFMA (hi/lo-split operands, 21-22 bits, f32 accumulate)",
            "data": "synthetic",
            "config": {
                "workload": f"ExactGP MLL evaluation (mBCG + SLQ), {'4-task Kronecker multitask RBF (x) index kernel' if T > 1 else kind}, n={n}, d={d}, "
                            f"{t_total} probes in total ({t} on rank 0) + y column, fused K*V HIP kernel ({args.contraction} contraction), "
                            + (f"rank-{precond_rank} pivoted-Cholesky preconditioner (built inside the step)" if precond_rank else "no preconditioner")
                            + f", cg_tolerance=1.0 (--config {args.config}: "
                            + {"metric": "the configuration BASELINE.json's metric is quoted on", "c2": "BASELINE configs[1]",
                               "c3": "BASELINE configs[2]", "c4": "BASELINE configs[3], 256 probes split over the ranks",
                               "c5": "BASELINE configs[4], 64 probes split over the ranks, one fused launch with 4 x columns per Kronecker MVM"}[args.config] + ")",
                "name": args.config, "kind": kind, "tasks": T, "n": n, "d": d, "probes_total": t_total, "probes_rank0": t, "rhs_columns_rank0": cols,
                "cg_iterations_per_step": iters_total / args.steps,
                "cg_iterations_all": iters_all,
                "parallelism": (f"probe-sharded x{world}, y column on rank 0" if R_ == 1 else
                                f"{P_} probe share(s) x {R_} row blocks (layout {'from --grid' if args.grid else 'chosen by distributed.choose_grid'}), "
                                "y column on the first share"),
                "grid": [P_, R_],
            },
            "mll": float(mll),
            "roofline": roofline,
            "contraction_note": ("timed steps and roofline: the fp32-MFMA contraction (v_mfma_f32_32x32x2_f32) -- the path BASELINE's metric names, NOT the "
                                 "library default; users get the split contraction (block 'split_contraction': ~2.6 x faster at the same 2e-5 bound) unless "
                                 "they ask for settings.split_contraction(False)") if args.contraction == "f32" else
                                "timed steps: the library-default split contraction (f32 emulated on the f16 matrix pipe)",
        }
        if parity is not None:
            out["parity"] = parity
        if other is not None:
            out["split_contraction" if args.contraction != "split" else "f32_contraction"] = other
        if extras is not None:
            out["extras"] = extras
        if not args.skip_cpu_baseline and world == 1:  # timed on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(n, d, cols * T, ls, kind=kind)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()   # rank 0 may still be in the (untimed) parity check
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
