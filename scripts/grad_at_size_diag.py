"""Where does the fused bilinear derivative (gpamd_kv_grad2_f32) deviate at size?  Per 128-row block, against float64 on the device.

For each (kind, d, n) and each contraction of W = L^T R (fp32 MFMA / hi-lo split f16) the kernel's per-unit partial sums (its float64
workspace) are summed over the j chunks of every 128-row block and compared with float64 torch on the same prepared points:

    g0[rb]   = sum_{i in rb} sum_j W_ij K_ij                     (W = L^T R in float64 from the float32 vectors)
    g1[rb,q] = sum_{i in rb} sum_j W_ij dk/ds_ij (z_iq - z_jq)^2

Truth for all n rows up to `full_rows` (default: every row below n = 131 072, a 16 384-row sample -- first, last and random 128-row
blocks -- above), so a deviation can be located: uniform over blocks (arithmetic) or concentrated (hazard / race).  Three repetitions per
contraction show whether it is deterministic.
Usage: python scripts/grad_at_size_diag.py [tag] [n ...]   -> gpurun_out/grad_at_size_diag_<tag>.json"""
import ctypes as C
import json
import math
import os
import sys

import torch

sys.path.insert(0, ".")
from gpytorch_amd import backend as B  # noqa: E402
from gpytorch_amd._lib import check, lib  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "x"
ns = [int(a) for a in sys.argv[2:]] or [16384, 131072, 500000]
dev = torch.device("cuda:0")
T = 65
LN2 = math.log(2.0)


def grad2_raw(xp, lt, rt, iso, split, want_gz1=False):
    """backend.kv_grad2 without the folding: returns (out [2 + dp] float32, per-row-block partials [nrb, 2 + dp] float64)."""
    L = lib()
    n, m, t = xp.n, xp.n, lt.shape[0]
    nd = int(L.gpamd_kv_grad2_workspace_doubles(n, m, t, xp.d))
    ws = torch.zeros(nd, device=dev, dtype=torch.float64)
    out = torch.empty(2 + xp.dp, device=dev, dtype=torch.float32)
    ns_ = int(L.gpamd_kv_grad2_split_workspace_floats(n, m)) if split else 0
    sws = torch.empty(ns_, device=dev, dtype=torch.float32) if split else None
    check(
        L.gpamd_kv_grad2_f32(
            *B.kind_args(xp), B._ptr(xp.xp), n, B._ptr(xp.xp), m, xp.d, None, B._ptr(lt), lt.stride(0), B._ptr(rt), rt.stride(0), t,
            1 if iso else 0, B._ptr(out), None, B.round_up(n, 4), B._ptr(ws), nd, None, 0, B.KV_SPLIT if split else 0, B._ptr(sws), ns_,
            B._stream(dev),
        ),
        "kv_grad2",
    )
    torch.cuda.synchronize()
    nrb = (n + 127) // 128
    nq = 2 + xp.dp
    units = nd // nq
    groups = 1 if (split or t <= 66) else None
    assert groups == 1
    S = units // nrb
    per = ws[: S * nrb * nq].view(S, nrb, nq).sum(0)   # unit = s * nrb + rb
    return out.double().cpu(), per.cpu()


def truth_blocks(kind, z, lt, rt, blocks):
    """float64 per-row-block sums for the given 128-row block indices: [len(blocks), 1 + d] (g0, g1_q)."""
    n, d = z.shape
    z64 = z.double()
    r64 = rt[:, :n].double()
    nn = (z64 * z64).sum(-1)
    out = torch.zeros(len(blocks), 1 + d, dtype=torch.float64, device=dev)
    CH = 16   # row blocks per chunk (2048 rows x n float64 = 8 GB at n = 5e5)
    for c0 in range(0, len(blocks), CH):
        bl = blocks[c0 : c0 + CH]
        rows = torch.cat([torch.arange(b * 128, min(n, b * 128 + 128), device=dev) for b in bl])
        zi = z64[rows]
        W = lt[:, rows].double().t() @ r64                                  # [rows, n]
        S = (nn[rows].unsqueeze(1) + nn.unsqueeze(0) - 2.0 * (zi @ z64.t())).clamp_min_(0.0)
        if kind == "rbf":
            K = torch.exp2(-S)
            dk = -LN2 * K
        else:   # matern52 on prepared coordinates: r = sqrt(S), k = (1 + r + S / 3) e^-r, dk/dS = -(1 + r) e^-r / 6
            r = S.sqrt()
            e = torch.exp(-r)
            K = (1.0 + r + S / 3.0) * e
            dk = -(1.0 + r) * e / 6.0
            del r, e
        A = W * dk
        g0 = (W * K).sum(1)
        del W, K, dk, S
        rs = A.sum(1)
        u = A @ z64                      # [rows, d]
        v = A @ (z64 * z64)
        gq = zi * zi * rs.unsqueeze(1) - 2.0 * zi * u + v
        del A
        # fold rows -> blocks
        owner = torch.repeat_interleave(torch.arange(len(bl), device=dev), torch.tensor([min(n, b * 128 + 128) - b * 128 for b in bl], device=dev))
        out[c0 : c0 + len(bl), 0].index_add_(0, owner, g0)
        out[c0 : c0 + len(bl), 1:].index_add_(0, owner, gq)
    return out.cpu()


results = []
for kind, d, ls in (("rbf", 3, 0.25), ("matern52", 10, 0.8)):
    for n in ns:
        X = torch.rand(n, d, generator=torch.Generator().manual_seed(0)).to(dev)
        xp = B.prep_points(kind, X, torch.tensor(ls), X.mean(0))
        g = torch.Generator(device=dev).manual_seed(7)
        lt = torch.randn(T, B.round_up(n, 4), device=dev, generator=g).abs_()
        rt = torch.randn(T, B.round_up(n, 4), device=dev, generator=g).abs_()
        nrb = (n + 127) // 128
        if n <= 131072:
            blocks = list(range(nrb))
        else:
            pick = torch.randperm(nrb, generator=torch.Generator().manual_seed(3))[:96].tolist()
            blocks = sorted(set(list(range(16)) + list(range(nrb - 16, nrb)) + pick))
        tr = truth_blocks(kind, xp.xp[:, :d], lt, rt, blocks)
        bi = torch.tensor(blocks)
        rec = dict(kind=kind, d=d, n=n, t=T, truth_blocks=len(blocks), library=os.environ.get("GPAMD_LIBRARY", "product"))
        for split in (False, True):
            for iso in (True, False):
                if (not iso) and d > 6 and split:
                    continue   # the library keeps ARD at d >= 8 on the fp32 contraction
                reps = []
                for rep in range(3):
                    out, per = grad2_raw(xp, lt, rt, iso, split)
                    p = per[bi]
                    e0 = ((p[:, 0] - tr[:, 0]) / tr[:, 0]).abs()
                    if iso:
                        t1 = tr[:, 1:].sum(1)
                        e1 = ((p[:, 1] - t1) / t1).abs()
                    else:
                        e1 = ((p[:, 1 : 1 + d] - tr[:, 1:]) / tr[:, 1:]).abs().max(1).values
                    tot0 = float(abs(p[:, 0].sum() - tr[:, 0].sum()) / tr[:, 0].sum().abs())
                    tot1 = float(abs((p[:, 1].sum() if iso else p[:, 1 : 1 + d].sum()) - tr[:, 1:].sum()) / tr[:, 1:].sum().abs())
                    reps.append(dict(g0_block_max=float(e0.max()), g0_block_median=float(e0.median()), g1_block_max=float(e1.max()),
                                     g1_block_median=float(e1.median()), g0_sample_total=tot0, g1_sample_total=tot1,
                                     bad_blocks_1e4=int(((e0 > 1e-4) | (e1 > 1e-4)).sum()), out1=float(out[1])))
                rec["split" if split else "fp32", "iso" if iso else "ard"] = reps
                print(kind, n, "split" if split else "fp32 ", "iso" if iso else "ard", " | ".join(
                    "g0 max %.1e med %.1e g1 max %.1e med %.1e bad %d" % (r["g0_block_max"], r["g0_block_median"], r["g1_block_max"],
                                                                         r["g1_block_median"], r["bad_blocks_1e4"]) for r in reps), flush=True)
        results.append({(k if isinstance(k, str) else "_".join(k)): v for k, v in rec.items()})
        del X, xp, lt, rt
        torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(results, open(f"gpurun_out/grad_at_size_diag_{tag}.json", "w"), indent=1)
