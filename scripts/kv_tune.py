#!/usr/bin/env python3
"""A/B the structural variants of the fused K*V kernel (csrc/tune/kv_mfma_tune.hpp, built into libgpamd_tune.so by `make -C gpytorch_amd/csrc tune`) in ONE process.

For each variant and a sweep of split counts S: 2 warm-up + R timed launches bracketed by HIP events
on the launch stream; reports median TFLOP/s (2 n^2 t / time) and the max abs deviation of the
result from variant 0.  Usage: python scripts/kv_tune.py [n] [rounds]
"""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpytorch_amd import backend as B  # noqa: E402
TUNE_LIB = os.path.join(ROOT, "gpytorch_amd", "csrc", "libgpamd_tune.so")  # make -C gpytorch_amd/csrc tune

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 6
t, d = 65, 3
dev = torch.device("cuda:0")
h = C.CDLL(TUNE_LIB)
h.gpamd_kv_partials_variant_f32.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int,
                                            C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
torch.manual_seed(0)
X = torch.rand(n, d, device=dev)
xp = B.prep_points("rbf", X, torch.tensor(0.25), X.mean(0))
ld = B.round_up(n, 4)
V = torch.randn(t, ld, device=dev)
V[:, n:] = 0
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
nv = h.gpamd_kv_variant_count()
ref = None
rows = []
only = [int(a) for a in sys.argv[3].split(',')] if len(sys.argv) > 3 else list(range(nv))
for v in only:
    bm, bn = C.c_int(0), C.c_int(0)
    h.gpamd_kv_variant_info(v, C.byref(bm), C.byref(bn))
    best = None
    for S in (7, 10, 12, 13, 17, 19, 23):
        jc = ((n + S - 1) // S + bn.value - 1) // bn.value * bn.value
        Se = (n + jc - 1) // jc
        P = torch.zeros(Se, t, ld, device=dev)
        times = []
        for it in range(2 + R):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = h.gpamd_kv_partials_variant_f32(v, xp.xp.data_ptr(), n, xp.xp.data_ptr(), n, V.data_ptr(), ld, t, P.data_ptr(), ld, Se, jc, st)
            e1.record()
            assert rc == 0, rc
            torch.cuda.synchronize()
            if it >= 2:
                times.append(e0.elapsed_time(e1))
        times.sort()
        med = times[len(times) // 2]
        tf = 2.0 * n * n * t / (med * 1e-3) / 1e12
        if best is None or tf > best[0]:
            out = P.sum(0)
            best = (tf, Se, med, times[0], out)
    out = best[4]
    if ref is None:
        ref = out
    dev_max = float((out - ref).abs().max())
    scale = float(ref.abs().max())
    rows.append(dict(variant=v, bm=bm.value, bn=bn.value, best_S=best[1], tflops=round(best[0], 2), ms_med=round(best[2], 3),
                     ms_min=round(best[3], 3), max_abs_dev_vs_v0=dev_max, ref_scale=scale))
    print(json.dumps(rows[-1]), flush=True)
print(json.dumps({"n": n, "t": t, "best": max(rows, key=lambda r: r["tflops"])}))
