"""Ablation timing of the split-operand kernel (libgpamd_tune.so, kv_gramh.hpp template parameter ABL): which part of the loop
bounds it?  python scripts/kgh_ablate.py [tag] [n] -> gpurun_out/kgh_ablate_<tag>.json"""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpytorch_amd import backend as B  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "x"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
cases = [tuple(int(v) for v in c.split(":")) for c in sys.argv[3].split(",")] if len(sys.argv) > 3 else \
    [(0, 2), (1, 2), (3, 2), (4, 2)]   # (ablation, row tiles per wave[, extra column 0 / 1])
dev = torch.device("cuda:0")
h = C.CDLL(os.path.join(ROOT, "gpytorch_amd", "csrc", "libgpamd_tune.so"))
f = h.gpamd_tune_kv_gramh_rbf3
f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
              C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
torch.manual_seed(0)
X = torch.rand(n, 3, device=dev)
xp = B.prep_points("rbf", X, torch.tensor(0.25), X.mean(0))
ld = B.round_up(n, 4)
ldh = (n + 127) // 128 * 128
V = torch.randn(65, ld, device=dev)
Vh = torch.randn(64, ldh, device=dev).half()
Vl = (1e-3 * torch.randn(64, ldh, device=dev)).half()
colmul = torch.ones(80, device=dev)
S, jc, _ = B.kv_plan("rbf", n, n, 3, 64, B.KV_GRAM | B.KV_SPLIT, ld)
P = torch.empty(S * 65 * ld, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
names = {0: "full", 1: "no generation VALU", 2: "no contraction MFMAs", 3: "V planes staged once", 4: "staged once, no barriers", 5: "A operands from one block", 6: "next tile prefetched into registers", 7: "one wave per SIMD", 8: "no sched_barrier pinning", 10: "Gram MFMA one step further ahead",
         110: "eight waves per workgroup", 120: "three waves per SIMD, one row tile per wave", 130: "eight waves per workgroup, four row tiles"}
out = []
for case in cases:
    abl, ni = case[0], case[1]
    ex = case[2] if len(case) > 2 else 0

    def run():
        rc = f(abl, ni, ex, xp.xp.data_ptr(), n, xp.xp.data_ptr(), n, V.data_ptr(), ld, Vh.data_ptr(), Vl.data_ptr(), ldh, colmul.data_ptr(),
               P.data_ptr(), ld, S, jc, st)
        assert rc == 0, rc
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        run()
    e1.record(); torch.cuda.synchronize()
    rec = dict(abl=abl, what=names[abl], ni=ni, extra_column=ex, S=S, ms=e0.elapsed_time(e1) / 3)
    res = P[: S * (64 + ex) * ld].view(S, 64 + ex, ld).sum(0)
    if abl == 0 and ni == 2:
        ref0 = {**globals().get("ref0", {}), ex: res.clone()}
    elif ex in globals().get("ref0", {}):
        rec["max_rel_dev_vs_full"] = float((res - ref0[ex]).abs().max() / ref0[ex].abs().max())
    print(json.dumps(rec), flush=True)
    out.append(rec)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/kgh_ablate_{tag}.json", "w"), indent=1)
