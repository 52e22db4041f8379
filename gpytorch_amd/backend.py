"""Thin torch-tensor wrappers over the C ABI (device pointers + the current HIP stream).

PyTorch is used here only for device memory, streams and tiny glue; every hot operation is a HIP
kernel in ``csrc/``.  All functions raise if the tensors are not on a ROCm device.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from ._lib import check, lib

KIND_IDS = {"rbf": 0, "matern12": 1, "matern32": 2, "matern52": 3}
NU_TO_KIND = {0.5: "matern12", 1.5: "matern32", 2.5: "matern52"}
MAX_INPUT_DIM = 16


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require_gpu(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"gpytorch_amd: `{name}` must live on a ROCm device (got {t.device}); the MI355X path has no CPU fallback"
        )


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def padded_dim(d: int) -> int:
    if d > MAX_INPUT_DIM:
        raise NotImplementedError(f"gpytorch_amd fused kernels support input dimension <= {MAX_INPUT_DIM} (got {d})")
    return round_up(d, 4)


KV_GRAM = 1              # flag of gpamd_kv_partials_f32 (include/gpamd.h)
GRAM_MAX_SQNORM = 32.0   # max |z|^2 for which the quadratic expansion keeps K within 1e-5 (kv_gram.hpp)
FORCE_KV_FLAGS = None    # tests / tuning: force 0 (direct-difference kernel) or KV_GRAM regardless of |z|


class PreparedPoints:
    """A point cloud converted for the fused kernels: float32 [n, dp], scaled by 1/lengthscale."""

    __slots__ = ("xp", "n", "d", "dp", "kind", "_zmax2")

    def __init__(self, xp, n, d, dp, kind):
        self.xp, self.n, self.d, self.dp, self.kind = xp, n, d, dp, kind
        self._zmax2 = None

    @property
    def zmax2(self) -> float:
        """max_i |z_i|^2 of the prepared (scaled, centred) points; one device reduction + sync, cached."""
        if self._zmax2 is None:
            self._zmax2 = float(self.xp.pow(2).sum(-1).max().item())
        return self._zmax2


def kv_flags(x1: PreparedPoints, x2: PreparedPoints, t: int) -> int:
    """Select the Gram-form generation kernel when it is both applicable and accurate (see kv_gram.hpp)."""
    if FORCE_KV_FLAGS is not None:
        return FORCE_KV_FLAGS
    if t <= 8 or x1.kind == "matern12":
        return 0
    return KV_GRAM if max(x1.zmax2, x2.zmax2 if x2 is not x1 else 0.0) <= GRAM_MAX_SQNORM else 0


def prep_points(kind: str, x: torch.Tensor, lengthscale: torch.Tensor, shift: torch.Tensor | None = None) -> PreparedPoints:
    """x: [n, d]; lengthscale: 1 or d values (any shape); shift: d values or None."""
    _require_gpu(x, "x")
    n, d = x.shape[-2], x.shape[-1]
    dp = padded_dim(d)
    x = x.detach().to(torch.float32).contiguous()
    ls = lengthscale.detach().to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
    if ls.numel() not in (1, d):
        raise ValueError(f"lengthscale must have 1 or {d} elements, got {ls.numel()}")
    sh = None if shift is None else shift.detach().to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
    xp = torch.empty(n, dp, device=x.device, dtype=torch.float32)
    check(
        lib().gpamd_prep_points_f32(
            KIND_IDS[kind], _ptr(x), n, d, x.stride(0), _ptr(ls), ls.numel(), _ptr(sh), _ptr(xp), dp, _stream(x.device)
        ),
        "prep_points",
    )
    return PreparedPoints(xp, n, d, dp, kind)


def to_probe_major(rhs: torch.Tensor) -> torch.Tensor:
    """[n, t] (any strides/dtype) -> float32 [t, ld] with ld = round_up(n, 4), zero padded."""
    n, t = rhs.shape[-2], rhs.shape[-1]
    ld = round_up(n, 4)
    out = torch.zeros(t, ld, device=rhs.device, dtype=torch.float32)
    out[:, :n] = rhs.detach().t()
    return out


def from_probe_major(vt: torch.Tensor, n: int) -> torch.Tensor:
    """float32 [t, ld] -> [n, t] contiguous."""
    return vt[:, :n].t().contiguous()


_ws_cache: dict = {}


def workspace(device, nfloats: int) -> torch.Tensor:
    key = (device.type, device.index)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nfloats:
        buf = torch.empty(max(nfloats, 1 << 20), device=device, dtype=torch.float32)
        _ws_cache[key] = buf
    return buf


def kv_plan(kind: str, n: int, m: int, d: int, t: int, flags: int, ldo: int):
    """(S, jchunk, workspace floats) for the kernel variant that (kind, d, t, flags) selects."""
    S, jc, ws = C.c_int(0), C.c_int(0), C.c_int64(0)
    check(lib().gpamd_kv_plan(KIND_IDS[kind], n, m, d, t, flags, ldo, C.byref(S), C.byref(jc), C.byref(ws)), "kv_plan")
    return S.value, jc.value, ws.value


def kv(x1: PreparedPoints, x2: PreparedPoints, vt: torch.Tensor, scale=None, dscale=None, vd=None, out=None, dvec=None):
    """out[t, ld_n] = scale * k(x1, x2) @ V + (dscale + dvec) .* Vd in probe-major layout.

    vt: [t, ldv] with ldv >= m; scale/dscale: 1-element device tensors or None; dvec: float32 [>= n] or None."""
    _require_gpu(vt, "vt")
    assert x1.kind == x2.kind and x1.dp == x2.dp
    t, ldv = vt.shape
    n, m = x1.n, x2.n
    ldo = round_up(n, 4)
    if out is None:
        out = torch.empty(t, ldo, device=vt.device, dtype=torch.float32)
    flags = kv_flags(x1, x2, t)
    S, jc, wsn = kv_plan(x1.kind, n, m, x1.d, t, flags, ldo)
    ws = workspace(vt.device, wsn)
    st = _stream(vt.device)
    L = lib()
    check(
        L.gpamd_kv_partials_f32(
            KIND_IDS[x1.kind], _ptr(x1.xp), n, _ptr(x2.xp), m, x1.d, _ptr(vt), ldv, t, _ptr(ws), ldo, S, jc,
            flags, None, st
        ),
        "kv_partials",
    )
    check(
        L.gpamd_kv_reduce_f32(
            _ptr(ws), S, ldo, t, n, _ptr(scale), _ptr(dscale), _ptr(dvec), _ptr(vd), 0 if vd is None else vd.stride(0),
            _ptr(out), out.stride(0), None, st,
        ),
        "kv_reduce",
    )
    return out


def kernel_dense(x1: PreparedPoints, x2: PreparedPoints, scale=None) -> torch.Tensor:
    out = torch.empty(x1.n, x2.n, device=x1.xp.device, dtype=torch.float32)
    check(
        lib().gpamd_kernel_dense_f32(
            KIND_IDS[x1.kind], _ptr(x1.xp), x1.n, _ptr(x2.xp), x2.n, x1.dp, _ptr(scale), _ptr(out), out.stride(0),
            _stream(out.device),
        ),
        "kernel_dense",
    )
    return out


def kernel_rows(x1: PreparedPoints, rows: torch.Tensor, x2: PreparedPoints, scale=None) -> torch.Tensor:
    rows = rows.to(device=x1.xp.device, dtype=torch.int64).contiguous()
    out = torch.empty(rows.numel(), x2.n, device=x1.xp.device, dtype=torch.float32)
    check(
        lib().gpamd_kernel_rows_f32(
            KIND_IDS[x1.kind], _ptr(x1.xp), _ptr(rows), rows.numel(), _ptr(x2.xp), x2.n, x1.dp, _ptr(scale), _ptr(out),
            out.stride(0), _stream(out.device),
        ),
        "kernel_rows",
    )
    return out


def kernel_diag(x1: PreparedPoints, x2: PreparedPoints, scale=None) -> torch.Tensor:
    assert x1.n == x2.n
    out = torch.empty(x1.n, device=x1.xp.device, dtype=torch.float32)
    check(
        lib().gpamd_kernel_diag_f32(
            KIND_IDS[x1.kind], _ptr(x1.xp), _ptr(x2.xp), x1.n, x1.dp, _ptr(scale), _ptr(out), _stream(out.device)
        ),
        "kernel_diag",
    )
    return out


def coldot(a: torch.Tensor, b: torch.Tensor, n: int) -> torch.Tensor:
    """Per-row (probe-major) inner products over the first n entries: out[c] = <a[c,:n], b[c,:n]>."""
    t = a.shape[0]
    out = torch.empty(t, device=a.device, dtype=torch.float32)
    scratch = torch.empty(t * 256, device=a.device, dtype=torch.float32)
    check(lib().gpamd_coldot_f32(_ptr(a), _ptr(b), a.stride(0), n, t, _ptr(out), _ptr(scratch), _stream(a.device)), "coldot")
    return out


def pivoted_cholesky(xp: PreparedPoints, scale, rank: int, tol: float):
    """Returns (L [m, n] row-major view of the rank-m factor (L^T of the reference's n x m), pivots[m], m)."""
    n = xp.n
    rank = min(rank, n)
    dev = xp.xp.device
    ldl = round_up(n, 4)
    L = torch.zeros(rank, ldl, device=dev, dtype=torch.float32)
    piv = torch.zeros(rank, device=dev, dtype=torch.int64)
    fwork = torch.empty(n + 4, device=dev, dtype=torch.float32)
    iwork = torch.zeros(2 + 2 * n, device=dev, dtype=torch.int32)
    check(
        lib().gpamd_pivoted_cholesky_f32(
            KIND_IDS[xp.kind], _ptr(xp.xp), n, xp.dp, _ptr(scale), rank, float(tol), _ptr(L), ldl, _ptr(piv), _ptr(fwork),
            _ptr(iwork), _stream(dev),
        ),
        "pivoted_cholesky",
    )
    m = int(iwork[0].item())
    return L[:m, :n], piv[:m], m


def kv_grad(x1: PreparedPoints, x2: PreparedPoints, lt: torch.Tensor, rt: torch.Tensor, iso: bool = False) -> torch.Tensor:
    """Fused bilinear derivative with W = lt^T rt (lt: [t, ld_n] over x1, rt: [t, ld_m] over x2).

    Returns float32 [1 + dp]:  g[0] = sum_ij W_ij k_ij;  g[1+q] = sum_ij W_ij dk/ds_ij (z_iq - z_jq)^2
    where z are the PREPARED coordinates and s the squared prepared distance."""
    _require_gpu(lt, "left")
    assert x1.kind == x2.kind and x1.dp == x2.dp and lt.shape[0] == rt.shape[0]
    t = lt.shape[0]
    dev = lt.device
    nd = int(lib().gpamd_kv_grad_workspace_doubles(x1.n, x2.n, t, x1.dp))
    ws = torch.empty(nd, device=dev, dtype=torch.float64)
    out = torch.empty(1 + x1.dp, device=dev, dtype=torch.float32)
    check(
        lib().gpamd_kv_grad_f32(
            KIND_IDS[x1.kind], _ptr(x1.xp), x1.n, _ptr(x2.xp), x2.n, x1.dp, _ptr(lt), lt.stride(0), _ptr(rt), rt.stride(0),
            t, 1 if iso else 0, _ptr(out), _ptr(ws), nd, _stream(dev),
        ),
        "kv_grad",
    )
    return out


# d s / d l factors: s = sum_q z_q^2-differences with z = coef * x / l  =>  ds_q/dl_q = -2 s_q / l_q
RBF_PREP_COEF = math.sqrt(0.5 * 1.4426950408889634)
