"""Few-column products over FEW output rows against many contracted points (the medium / wide row regions of a block-centred product; test points
against the training set): the few-column kernels (kv_valu / kv_gramv) against the split kernels at one 32-column tile (GPAMD_KV_SPLIT_FEW).
python scripts/region_few_timing.py -> gpurun_out/region_few_timing.json"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, ".")
from gpytorch_amd import backend as B  # noqa: E402
from gpytorch_amd._lib import check, lib  # noqa: E402

dev = torch.device("cuda:0")
L = lib()
out = []
m, d = 217_437, 3
g = torch.Generator(device=dev).manual_seed(0)
X2 = torch.rand(m, d, device=dev, generator=g)
for kind in ("matern52", "rbf"):
    x2 = B.prep_points(kind, X2, torch.tensor(0.3), X2.mean(0))
    for n in (1024, 4096, 8192, 16384, 32768, 65536):
        x1 = B.PreparedPoints(x2.xp[:n].contiguous(), n, d, x2.dp, kind, None)
        for t in (1, 2, 4):
            vt = torch.randn(t, B.round_up(m, 4), device=dev, generator=g)
            rec = dict(kind=kind, n=n, m=m, t=t)
            ref = None
            for name, flags in (("direct_few_column_kernel", B.KV_SPLIT), ("direct_split_few", B.KV_SPLIT | B.KV_SPLIT_FEW),
                                ("gram_few_column_kernel", B.KV_GRAM | B.KV_SPLIT), ("gram_split_few", B.KV_GRAM | B.KV_SPLIT | B.KV_SPLIT_FEW)):
                ldo = B.round_up(n, 4)
                S, jc, wsn = B.kv_plan(kind, n, m, d, t, flags, ldo)
                P = B.workspace(dev, wsn)
                res = torch.empty(t, ldo, device=dev)
                st = B._stream(dev)

                def run():
                    check(L.gpamd_kv_partials_f32(*B.kind_args(x1), B._ptr(x1.xp), n, B._ptr(x2.xp), m, d, None, B._ptr(vt), vt.stride(0), t, B._ptr(P), ldo, S, jc, flags, None, st), "kv")
                    check(L.gpamd_kv_reduce_f32(B._ptr(P), S, ldo, t, n, None, None, None, None, 0, B._ptr(res), ldo, None, st), "reduce")

                run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run()
                e1.record()
                torch.cuda.synchronize()
                rec[name + "_ms"] = e0.elapsed_time(e1) / 10
                rec[name + "_S"] = S
                if ref is None:
                    ref = res.clone()
                else:
                    rec[name + "_rel_dev"] = float((res - ref).abs().max() / ref.abs().max())
            print(json.dumps(rec), flush=True)
            out.append(rec)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/region_few_timing.json", "w"), indent=1)
