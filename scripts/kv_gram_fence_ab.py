"""What the explicit wait states behind the Gram MFMAs of kv_gram_kernel (round 5) cost on the headline kernel: the SAME template in the tune
library as the product builds it (20 wait states + scheduling barrier), fully fenced (32), in round 4's form (the toolchain's 12 only) and with the wait
states but without the second scheduling barrier -- one box, one process, n = 500 000, RBF, d = 3, 65 columns, 6 launches each (HIP events), twice.
Usage: python scripts/kv_gram_fence_ab.py [out.json]"""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpytorch_amd import backend as B  # noqa: E402

dev = torch.device("cuda:0")
h = C.CDLL(os.path.join(ROOT, "gpytorch_amd", "csrc", "libgpamd_tune.so"))
f = h.gpamd_tune_hazard_launch
f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
              C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
n, d, t = 500_000, 3, 65
g = torch.Generator().manual_seed(0)
X = torch.rand(n, d, generator=g).to(dev)
xp = B.prep_points("rbf", X, torch.tensor(0.25), X.mean(0))
ld = B.round_up(n, 4)
V = torch.randn(t, ld, generator=g).to(dev)
S, jc, _ = B.kv_plan("rbf", n, n, d, t, B.KV_GRAM, ld)
P = torch.empty(S * t * ld, device=dev)
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
names = {0: "product (20 wait states + scheduling barrier)", 1: "fully fenced (32 wait states)", 2: "round 4's form (toolchain's 12 only)", 3: "wait states, no second scheduling barrier"}
out = {"n": n, "d": d, "t": t, "S": S, "rounds": []}
ref = None
for rnd in range(2):
    rec = {}
    for safe in (0, 2, 3, 1):
        def launch():
            rc = f(0, 2, safe, xp.xp.data_ptr(), n, xp.xp.data_ptr(), n, V.data_ptr(), ld, t, None, None, 0, None, P.data_ptr(), ld, S, jc, st)
            assert rc == 0, rc
        launch()
        torch.cuda.synchronize()
        if ref is None:
            ref = P.clone()
        same = bool(torch.equal(P, ref))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6):
            launch()
        e1.record()
        torch.cuda.synchronize()
        rec[names[safe]] = {"ms_per_launch": e0.elapsed_time(e1) / 6, "bitwise_equal_to_the_product_build": same}
    out["rounds"].append(rec)
    print(json.dumps(rec), flush=True)
path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/kv_gram_fence_ab.json"
os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
json.dump(out, open(path, "w"), indent=1)
