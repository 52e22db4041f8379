"""Fused K*V beyond 16 input dimensions against the row-block path it replaces (HIP events):
    python scripts/kv_highdim_timing.py [n] -> gpurun_out/kv_highdim_timing.json
For d in {20, 24, 32} and t in {1, 11, 65}: direct-difference kernels, Gram form with the fp32-MFMA contraction, Gram form with the split
contraction (the default), and the generic row-block path (HIP-generated dense row blocks x library GEMM, timed on a row sample and scaled)."""
import json
import os
import sys

import torch

sys.path.insert(0, ".")
from gpytorch_amd import backend as B  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
dev = torch.device("cuda:0")
torch.manual_seed(0)
PEAK = 157.3


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


rows = []
for kind, d in (("rbf", 20), ("rbf", 24), ("matern52", 24), ("rbf", 32)):
    X = torch.rand(n, d, device=dev)
    ls = 0.2 + 0.08 * d
    xp = B.prep_points(kind, X, torch.tensor(ls), X.mean(0))
    for t in (1, 11, 65):
        V = torch.randn(t, B.round_up(n, 4), device=dev)
        rec = {"kind": kind, "d": d, "n": n, "t": t}
        try:
            for name, flags in (("direct", 0), ("gram_fp32", B.KV_GRAM), ("gram_split", B.KV_GRAM | B.KV_SPLIT)):
                B.FORCE_KV_FLAGS = flags
                ms = timed(lambda: B.kv(xp, xp, V), 2)
                rec[name + "_ms"] = ms
                rec[name + "_tflops"] = 2.0 * n * n * t / ms / 1e9
        finally:
            B.FORCE_KV_FLAGS = None
        if t == 65:
            # the row-block path (what d > 16 took until round 4): timed on the first 16 384 rows, scaled to n rows
            ns = min(n, 16384)
            xs = B.PreparedPoints(xp.xp[:ns].contiguous(), ns, d, xp.dp, kind, xp.param)
            B.FORCE_GENERIC = True
            try:
                ms = timed(lambda: B.kv(xs, xp, V), 1)
            finally:
                B.FORCE_GENERIC = False
            rec["row_block_path_ms_scaled"] = ms * n / ns
            rec["speedup_default_vs_row_blocks"] = rec["row_block_path_ms_scaled"] / rec["gram_split_ms"]
        rec["frac_fp32_mfma_peak_gram_fp32"] = rec["gram_fp32_tflops"] / PEAK
        print(json.dumps(rec), flush=True)
        rows.append(rec)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/kv_highdim_timing.json", "w"), indent=1)
