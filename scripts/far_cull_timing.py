"""Far-pair tile culling (settings.far_pair_cutoff) on the 3droad-shaped cloud of scripts/reference_workloads.py: time of one fused K*V with and
without culling per lengthscale and column count, the share of (row block, tile) pairs that survive, and the deviation of the culled product from the
un-culled one against the stated bound eps * sum_j |V_jc|.   python scripts/far_cull_timing.py [n] -> gpurun_out/far_cull_timing.json"""
import json
import os
import sys
import warnings

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "scripts")
import gpytorch_amd as g  # noqa: E402
from gpytorch_amd import backend as B  # noqa: E402
from reference_workloads import road_like  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 217_437
dev = torch.device("cuda:0")
X, _ = road_like(n, 0)
Xd = X.to(dev)
out = []


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


warnings.simplefilter("ignore")
for kind in ("matern52", "rbf"):
    for ls in (0.03, 0.05, 0.1, 0.2, 0.35):
        xp = B.prep_points(kind, Xd, torch.tensor([ls]), Xd.mean(0))
        mode = B.gram_mode(xp, xp)
        sv = xp.sorted_view()
        for t in (11, 33, 65):
            V = torch.randn(t, B.round_up(n, 4), generator=torch.Generator().manual_seed(t)).to(dev)
            V[:, n:] = 0
            rec = dict(kind=kind, n=n, lengthscale=ls, t=t, gram_mode=mode, compact_rows=sv.n_compact if mode == 2 else None,
                       medium_rows=(sv.n_block - sv.n_compact) if mode == 2 else None, max_sq_scaled_radius=float(xp.zmax2))
            rec["ms_every_pair"] = timed(lambda: B.kv(xp, xp, V))
            ref = B.kv(xp, xp, V)
            for eps in (1e-7, 1e-5):
                with g.settings.far_pair_cutoff(eps):
                    sq = B.far_cull(xp, xp)
                    if sq is None:
                        rec[f"eps{eps:g}"] = None
                        continue
                    ms = timed(lambda: B.kv(xp, xp, V))
                    got = B.kv(xp, xp, V)
                    dev_abs = (got - ref)[:, :n].abs().amax(1)
                    bound = eps * V[:, :n].abs().sum(1)
                    rec[f"eps{eps:g}"] = dict(ms=ms, speedup=rec["ms_every_pair"] / ms, kept_fraction_512=B.far_kept_fraction(xp, xp, sq, 512),
                                              kept_fraction_128=B.far_kept_fraction(xp, xp, sq, 128), sq_cutoff=sq,
                                              max_dev_over_bound=float((dev_abs / bound).max()), max_dev_rel=float(dev_abs.max() / ref.abs().max()))
            print(json.dumps(rec), flush=True)
            out.append(rec)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/far_cull_timing.json", "w"), indent=1)
