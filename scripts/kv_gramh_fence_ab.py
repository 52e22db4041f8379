"""What a full data-dependent fence (32 wait states tied to the Gram result registers, common.hpp::mfma_result_fence) behind every Gram MFMA would cost
kv_gramh_kernel -- the library-default K*V (split contraction), which today reads its Gram results at the toolchain's 12 wait states with one contraction
MFMA among them (profiles/r05_final_isa_hazard_audit.json).  Same A/B as scripts/kv_gram_fence_ab.py recorded for kv_gram_kernel (0.26 %): the SAME template
twice in the tune library (SAFE = 0: as the product builds it, SAFE = 1: fully fenced), one box, one process, HIP events, bitwise comparison of the slabs.
Shapes: the headline split kernel (RBF, d = 3, n = 500 000, 65 columns) and C3's (Matern-5/2, d = 10, n = 500 000, 65 columns).
Usage: python scripts/kv_gramh_fence_ab.py [out.json]"""
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
out = {"cases": []}
for variant, kind, d, ls in ((2, "rbf", 3, 0.25), (3, "matern52", 10, 0.8)):
    n, t = 500_000, 65
    g = torch.Generator().manual_seed(variant)
    X = torch.rand(n, d, generator=g).to(dev)
    xp = B.prep_points(kind, X, torch.tensor(ls), X.mean(0))
    ld = B.round_up(n, 4)
    ldh = (n + 127) // 128 * 128
    V = torch.randn(t, ld, generator=g).to(dev)
    tc = 64
    Vh = (4096.0 * torch.randn(tc, ldh, generator=g)).to(dev).half()
    Vl = torch.randn(tc, ldh, generator=g).to(dev).half()
    colmul = torch.ones(tc + 1, device=dev)
    S, jc, _ = B.kv_plan(kind, n, n, d, t, B.KV_GRAM | B.KV_SPLIT, ld)
    P = torch.empty(S * t * ld, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    rec = {"kind": kind, "d": d, "n": n, "t": t, "S": S, "rounds": []}
    ref = None
    for rnd in range(3):
        row = {}
        for safe in (0, 2, 1):
            def launch():
                rc = f(1, variant, safe, xp.xp.data_ptr(), n, xp.xp.data_ptr(), n, V.data_ptr(), ld, t, Vh.data_ptr(), Vl.data_ptr(), ldh, colmul.data_ptr(),
                       P.data_ptr(), ld, S, jc, st)
                assert rc == 0, rc
            launch()
            torch.cuda.synchronize()
            if ref is None:
                ref = P.clone()
            same = bool(torch.equal(P, ref))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                launch()
            e1.record()
            torch.cuda.synchronize()
            row[{0: "product (8 tied wait states + toolchain's 12 + one contraction MFMA; round 6)", 1: "fully fenced (32 wait states behind every Gram MFMA)",
                 2: "round 5's form (toolchain's 12 + one contraction MFMA)"}[safe]] = {
                "ms_per_launch": e0.elapsed_time(e1) / 10, "bitwise_equal_to_the_product_build": same}
        rec["rounds"].append(row)
        print(kind, json.dumps(row), flush=True)
    out["cases"].append(rec)
    del X, V, Vh, Vl, P, ref
    torch.cuda.empty_cache()
path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/kv_gramh_fence_ab.json"
os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
json.dump(out, open(path, "w"), indent=1)
