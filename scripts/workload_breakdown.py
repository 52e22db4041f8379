"""Where the time goes on the reference's own workloads (road3d-shaped Adam iterations, protein-shaped L-BFGS closures;
scripts/reference_workloads.py): a host timeline by phase and the GPU-busy fraction.

    python scripts/workload_breakdown.py road3d|protein plain|phases [iters]

  plain   un-instrumented iterations, wall clock per iteration only.  Run it under `rocprofv3 --kernel-trace --stats`: the sum of the
          kernel durations over the sum of the timed iterations is the GPU-busy fraction (nothing else in the process launches kernels
          worth mentioning: the data is generated on the host).
  phases  the same iterations with a device synchronisation + perf_counter around every phase of one MLL evaluation (prepare points /
          Hilbert order / preconditioner build / probe draw / mBCG loop with its K*V launches and preconditioner applies / SLQ / backward
          kernel / optimiser), EXCLUSIVE times (a phase's own time, children subtracted).  Synchronising distorts overlap but not
          attribution: it says which host section the un-overlapped time of `plain` belongs to.
Output: gpurun_out/workload_breakdown_<workload>_<mode>.json"""
import json
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scripts.reference_workloads import _model, _path, gaussian_features, road_like  # noqa: E402

workload, mode = sys.argv[1], sys.argv[2]
iters = int(sys.argv[3]) if len(sys.argv) > 3 else (3 if workload == "road3d" else 5)
dev = torch.device("cuda:0")
torch.cuda.set_device(0)

import gpytorch_amd as g  # noqa: E402
from gpytorch_amd import backend as B, bbmm, functions, linear_cg as LCG  # noqa: E402

acc, stack = {}, []


def timed(label, fn):
    def wrapper(*a, **k):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        stack.append(0.0)
        try:
            return fn(*a, **k)
        finally:
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            child = stack.pop()
            e = acc.setdefault(label, [0.0, 0.0, 0])
            e[0] += dt - child
            e[1] += dt
            e[2] += 1
            if stack:
                stack[-1] += dt
    return wrapper


def patch(obj, name, label):
    setattr(obj, name, timed(label, getattr(obj, name)))


if mode == "phases":
    patch(functions, "_prep", "prepare points (x / l, centring)")
    patch(B.PreparedPoints, "sorted_view", "Hilbert order + block centres (SortedView)")
    patch(bbmm, "build_preconditioner", "preconditioner build (pivoted Cholesky + QR)")
    patch(bbmm, "probe_vectors", "probe draw")
    patch(bbmm, "linear_cg", "mBCG loop: host side, vector kernels, polls")
    patch(bbmm, "slq_logdet", "SLQ (eigh of the tridiagonals on the host)")
    patch(B, "kv_partials_sorted", "fused K*V launches")
    patch(LCG.Preconditioner, "apply_", "preconditioner apply")
    patch(functions, "hyper_grads", "backward: fused bilinear derivative (+ its operand preparation)")
    patch(B, "kv_grad2", "backward: kv_grad2 launches incl. split pre-pass")

if workload == "road3d":
    n = 217_437
    X, y = road_like(n, 0)
    m, lik = _model(g, "matern52", X, y, dev)
    stages = [("lengthscale 0.05 (start of training)", 0.05), ("lengthscale 0.1", 0.1), ("lengthscale 0.2 (later iterations)", 0.2)]
else:
    n = 36_584
    X, y = gaussian_features(n, 9, 0)
    m, lik = _model(g, "rbf", X, y, dev)
    stages = [("default initialisation, max_preconditioner_size(100), deterministic probes", None)]
m.train()
lik.train()
mll = g.ExactMarginalLogLikelihood(lik, m)
opt = torch.optim.Adam(m.parameters(), lr=0.01)
S = g.settings
out = {"workload": workload, "mode": mode, "n": n, "iterations_per_stage": iters, "stages": []}
ctx = (S.max_preconditioner_size(100), S.deterministic_probes(True)) if workload == "protein" else ()
for c in ctx:
    c.__enter__()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for label, ls in stages:
        if ls is not None:
            m.covar_module.base_kernel.lengthscale = ls
        walls, its = [], []
        for i in range(iters + 1):                      # the first evaluation of a stage is a warm-up (allocations, order cache)
            if i == 1:
                acc.clear()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            opt.zero_grad()
            tf = time.perf_counter()
            loss = -mll(m(m.train_inputs[0]), m.train_targets)
            if mode == "phases":
                torch.cuda.synchronize(dev)
            tb = time.perf_counter()
            loss.backward()
            if mode == "phases":
                torch.cuda.synchronize(dev)
            to = time.perf_counter()
            opt.step()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            if i:
                walls.append(t1 - t0)
                its.append(LCG.LAST_INFO.iterations)
                if mode == "phases":
                    for k_, v_ in (("forward total", tb - tf), ("backward total", to - tb), ("optimiser step", t1 - to)):
                        e = acc.setdefault(k_, [0.0, 0.0, 0])
                        e[0] += v_
                        e[1] += v_
                        e[2] += 1
        st = {"stage": label, "seconds_per_iteration": walls, "cg_iterations": its}
        pth = _path(m, lik)
        m.train(), lik.train()
        st["kernel_path"], st["max_sq_scaled_radius"], st["rows_by_region"] = pth[0], pth[1], getattr(_path, "last_regions", None)
        if mode == "phases":
            st["phases_exclusive_seconds_per_iteration"] = {k_: v_[0] / iters for k_, v_ in sorted(acc.items(), key=lambda kv: -kv[1][0])}
            st["phases_inclusive_seconds_per_iteration"] = {k_: v_[1] / iters for k_, v_ in acc.items()}
            st["calls_per_iteration"] = {k_: v_[2] / iters for k_, v_ in acc.items()}
        out["stages"].append(st)
out["timed_seconds_total"] = sum(sum(s["seconds_per_iteration"]) for s in out["stages"])
os.makedirs("gpurun_out", exist_ok=True)
with open(f"gpurun_out/workload_breakdown_{workload}_{mode}.json", "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
