"""A/B of the DMA-staged, software-pipelined Gram kernel (csrc/tune/kv_gram2.hpp, libgpamd_tune.so) against the product
kernel (kv_gram.hpp): equality of results on ragged shapes, then timing.  RBF, d <= 3, 33 <= t <= 65.
Build first: make -C gpytorch_amd/csrc tune.   Usage: python scripts/async_check.py"""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpytorch_amd import backend as B  # noqa: E402

h = C.CDLL(os.path.join(ROOT, "gpytorch_amd", "csrc", "libgpamd_tune.so"))
h.gpamd_tune_kv_gram2_rbf3.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64,
                                       C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")


def gram2(p1, p2, vt):
    t = vt.shape[0]
    ldo = B.round_up(p1.n, 4)
    S, jc, wsn = B.kv_plan("rbf", p1.n, p2.n, 3, t, B.KV_GRAM, ldo)
    P = torch.zeros(S, t, ldo, device=dev)
    rc = h.gpamd_tune_kv_gram2_rbf3(p1.xp.data_ptr(), p1.n, p2.xp.data_ptr(), p2.n, vt.data_ptr(), vt.stride(0), t, P.data_ptr(), ldo, S, jc,
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    return P.sum(0)


out = {"equal": [], "timing": []}
for n, m, d, t in [(700, 1100, 3, 33), (1025, 1300, 2, 64), (600, 2100, 3, 65), (257, 128, 1, 64), (5000, 4097, 3, 40), (300, 31, 3, 65)]:
    g = torch.Generator().manual_seed(n + m)
    X1, X2 = torch.rand(n, d, generator=g).to(dev), torch.rand(m, d, generator=g).to(dev)
    sh = X1.mean(0)
    p1, p2 = B.prep_points("rbf", X1, torch.tensor(0.4), sh), B.prep_points("rbf", X2, torch.tensor(0.4), sh)
    vt = torch.randn(t, B.round_up(m, 4), device=dev)
    vt[:, m:] = 0
    B.FORCE_KV_FLAGS = B.KV_GRAM
    a = B.kv(p1, p2, vt).clone()
    B.FORCE_KV_FLAGS = None
    b = gram2(p1, p2, vt)
    out["equal"].append(dict(n=n, m=m, d=d, t=t, max_rel=float((a[:, :n] - b[:, :n]).abs().max() / a[:, :n].abs().max())))
for n in (100_000, 500_000):
    X = torch.rand(n, 3, device=dev)
    xp = B.prep_points("rbf", X, torch.tensor(0.25), X.mean(0))
    vt = torch.randn(65, B.round_up(n, 4), device=dev)
    row = dict(n=n)
    for name, fn in (("sync", lambda: B.kv(xp, xp, vt)), ("dma_pipelined", lambda: gram2(xp, xp, vt))):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            fn()
        e1.record()
        torch.cuda.synchronize()
        row[name + "_tflops"] = 2.0 * n * n * 65 / (e0.elapsed_time(e1) / 3 * 1e-3) / 1e12
    out["timing"].append(row)
print(json.dumps(out))
