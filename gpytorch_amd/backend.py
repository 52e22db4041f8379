"""Thin torch-tensor wrappers over the C ABI (device pointers + the current HIP stream).

PyTorch is used here only for device memory, streams and tiny glue; every hot operation is a HIP
kernel in ``csrc/``.  All functions raise if the tensors are not on a ROCm device.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import torch

from ._lib import check, lib

KIND_IDS = {"rbf": 0, "matern12": 1, "matern32": 2, "matern52": 3, "rq": 4}
NU_TO_KIND = {0.5: "matern12", 1.5: "matern32", 2.5: "matern52"}
MAX_INPUT_DIM = 32          # fused float32 kernels: 1 .. 32 input dimensions (csrc/kv_dispatch.hpp KV_MAX_DIM; 16 until round 5)
MAX_GRAD2_ARD_DIM = 16      # per-dimension sums / input gradients of the Gram-form derivative kernel (kv_grad2.hpp MODE 1) exist up to here


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
    """Row stride of a prepared cloud: 4 * ceil(d / 4), except 25 .. 32 dimensions, which share the D = 32 kernels (stride 32)."""
    return 32 if 24 < d <= 32 else round_up(d, 4)


def work_dtype(t: torch.Tensor) -> torch.dtype:
    """float64 tensors are computed in float64 (generic path), everything else in float32 (fused path)."""
    return torch.float64 if t.dtype == torch.float64 else torch.float32


KV_GRAM = 1              # flag of gpamd_kv_partials_f32 (include/gpamd.h)
KV_WIDE = 2              # with KV_GRAM, tuning / A-B only: keep 3..32 columns off the 4-column-group kernel (kv_gram4.hpp)
KV_G4 = 4                # with KV_GRAM, tuning / A-B only: 9..12 columns on kv_gram4 (three groups) instead of kv_gram16
KV_SPLIT = 8             # with KV_GRAM, >= 5 columns: contraction of hi/lo-split operands on the f16 matrix pipe (kv_gramh.hpp)
SPLIT_CONTRACTION = None  # tests / tuning: force True / False; None -> settings.split_contraction
GRAM_MAX_SQNORM = 32.0   # max |z|^2 for which the split-f16 quadratic expansion keeps K within 2e-5 (gram_f16.hpp)
# block-centred expansion: error of S ~ 2^-22 (sqrt(S) + 2 r)^2 with r the radius of the workgroup's row block: blocks within the radius the
# cloud-centred rule accepts for EVERY point (relative error <= 2e-5 in K, worst case) run on the Gram-form kernels, the rows of wider
# blocks (sparse tails) on the direct-difference kernel; beyond GRAM_MAX_WIDE_FRACTION of the rows the whole product falls back
GRAM_MAX_BLOCK_SQRADIUS = 32.0
GRAM_MAX_WIDE_FRACTION = 0.8    # (0.25 until round 5: any row off the direct-difference kernels is a gain -- the Hilbert order is cached per cloud)
# block-centred expansion: (max |z1| + max |z2|)^2.  The split norm |z_j - c|^2 of a contracted point saturates at 60000 (f16 range,
# csrc/gram_f16.hpp gram_norm_clamp): invisible for the families that are zero to f32 precision at such distances as long as |z_j - c| <= 5000;
# the heavy-tailed RQ keeps the unsaturated range
GRAM_MAX_EXTENT_SQ = 2.5e7
GRAM_MAX_EXTENT_SQ_RQ = 60000.0
DIRECT_SPLIT_MAX_DIM = 10   # csrc/kv_directh.hpp KDH_MAX_DIM
KV_BLOCK128 = 16         # flag of gpamd_kv_partials_f32: the caller bounds 128-row blocks only (SortedView's medium groups)
KV_SPLIT_FEW = 32        # with KV_SPLIT: fewer than five columns on the split kernels too (far-pair culling: only they walk tile lists)
_warned_fallback = set()
FORCE_GENERIC = False    # tests: send float32 / d <= 16 problems down the generic (row-block + GEMM) path too
FORCE_KV_FLAGS = None    # tests / tuning: force 0 (direct-difference kernel) or KV_GRAM regardless of |z|


class PreparedPoints:
    """A point cloud converted for the fused kernels: float32 [n, dp], scaled by 1/lengthscale."""

    __slots__ = ("xp", "n", "d", "dp", "kind", "_zmax2", "param", "_sorted", "order_key", "_far_keep")

    def __init__(self, xp, n, d, dp, kind, param=None):
        self.xp, self.n, self.d, self.dp, self.kind = xp, n, d, dp, kind
        self._zmax2 = None
        self.param = param   # shape parameter of the covariance family (RQ: alpha, a Python float) or None
        self._far_keep = {}  # far-pair culling: surviving share of (512-row block, tile) pairs per (contracted cloud, cutoff), far_kept_fraction
        self._sorted = None  # lazily: SortedView (Hilbert order + chunk centres) for the block-centred Gram expansion
        self.order_key = None  # identity of the SOURCE cloud when the scaling is uniform (prep_points): the Hilbert order is then shared
        #                        by every evaluation of a training run (it is invariant under translation and uniform scaling)

    def sorted_view(self):
        """The rows of ``xp`` along a Hilbert curve, with the centre of every 128-row chunk and the largest block radius: what the
        Gram-form kernels need once the cloud as a whole is too wide for the cloud-centred expansion (max |z|^2 > 32).  Computed once
        per prepared cloud (a dozen elementwise passes + one sort of n keys)."""
        if self._sorted is None:
            self._sorted = SortedView(self)
        return self._sorted

    @property
    def dtype(self):
        return self.xp.dtype

    @property
    def fused(self) -> bool:
        """True when the fused float32 MFMA / VALU kernels apply (float32, d <= 32); otherwise products go through
        the generic path: HIP-generated dense row blocks of K times V with a library GEMM (``kv_chunked``)."""
        return self.xp.dtype == torch.float32 and self.d <= MAX_INPUT_DIM and not FORCE_GENERIC

    @property
    def zmax2(self) -> float:
        """max_i |z_i|^2 of the prepared (scaled, centred) points; one device reduction + sync, cached."""
        if self._zmax2 is None:
            self._zmax2 = float(self.xp.pow(2).sum(-1).max().item())
        return self._zmax2


def hilbert_order(z: torch.Tensor, bits: int | None = None) -> torch.Tensor:
    """Permutation that sorts the rows of ``z`` [n, d] along a Hilbert curve (Skilling, "Programming the Hilbert curve", AIP Conf. Proc.
    707, 2004: axes -> transposed index with O(bits * d) elementwise integer passes, then one sort of n keys).  Unlike the Z-order
    (Hilbert) curve the Hilbert curve has no jumps -- consecutive cells are face neighbours -- so EVERY run of consecutive points is
    spatially compact, which is what the block-centred Gram expansion needs."""
    n, d = z.shape
    if bits is None:
        bits = max(1, min(16, 62 // d))
    lo = z.min(0).values
    span = (z.max(0).values - lo).max().clamp_min(1e-30)
    top = 2 ** bits - 1
    X = [((z[:, k] - lo[k]) / span * top).round().to(torch.int64).clamp_(0, top) for k in range(d)]
    Q = 1 << (bits - 1)
    while Q > 1:                                    # inverse undo
        P = Q - 1
        for i in range(d):
            hit = (X[i] & Q) != 0
            t = (X[0] ^ X[i]) & P
            x0_inv = X[0] ^ P                       # invert low bits of X[0]
            x0_exc = X[0] ^ t                       # exchange low bits of X[0] and X[i]
            xi_exc = X[i] ^ t
            X[0] = torch.where(hit, x0_inv, x0_exc)
            if i:
                X[i] = torch.where(hit, X[i], xi_exc)
        Q >>= 1
    for i in range(1, d):                           # Gray encode
        X[i] = X[i] ^ X[i - 1]
    t = torch.zeros_like(X[0])
    Q = 1 << (bits - 1)
    while Q > 1:
        t = torch.where((X[d - 1] & Q) != 0, t ^ (Q - 1), t)
        Q >>= 1
    X = [x ^ t for x in X]
    key = torch.zeros_like(X[0])
    for b in range(bits - 1, -1, -1):               # interleave: bit b of X[0], X[1], ..., then bit b - 1 ...
        for k in range(d):
            key = (key << 1) | ((X[k] >> b) & 1)
    return torch.argsort(key)


# Hilbert permutations by source cloud (data pointer, version counter, shape, device): hyper-parameters change every training step and the
# prepared points with them, but for a single lengthscale the ORDER of the rows along the curve does not (any permutation is valid for the
# kernels -- compactness is re-evaluated on the actual coordinates every time -- so a stale entry can only cost speed, never accuracy)
_ORDER_CACHE: dict = {}


class SortedView:
    """Hilbert-sorted copy of a prepared cloud for the block-centred Gram expansion (csrc/gram_f16.hpp ``load_center``).

    Rows are grouped in runs of 512 consecutive points of the Hilbert order (the largest row block of any Gram-form kernel).  A group is
    COMPACT when every 128 / 256 / 512-row block inside it stays within ``GRAM_MAX_BLOCK_SQRADIUS`` of its own centre -- the radius
    the cloud-centred rule accepts for every point, i.e. <= 2e-5 relative error in K.  The sparse tails of a cloud (outlying points many
    lengthscales from everything: Gaussian inputs, short lengthscales) form WIDE groups whatever the order; they are moved behind the
    compact ones and their rows are produced by the direct-difference kernels in a further launch into the same slabs (1.1-1.6x slower per row,
    a few percent of the rows), so the accuracy bound holds for every entry.

    ``perm``: sorted row k is original row perm[k];  ``inv_pad``: [round_up(n, 4)] gather index that takes a probe-major row in sorted
    order back to the original order (identity on the padding);  ``centers``: [ceil(n / 128), dp] chunk means of the sorted rows;
    ``n_compact``: rows [0, n_compact) are the compact groups;  ``n_block``: rows [n_compact, n_block) are the MEDIUM chunks -- 128-row blocks
    within the policy whose 256 / 512-row blocks are not (elongated runs: points along a curve, short lengthscales): the kernels that centre
    128-row blocks serve them (the split kernel at one row tile per wave, flag GPAMD_KV_BLOCK128; the derivative kernel always centres 128-row blocks),
    rows [n_block, n) are the wide groups;  ``r2``: the largest admitted block radius^2."""

    def __init__(self, x: "PreparedPoints"):
        n, dp = x.n, x.dp
        dev = x.xp.device
        perm = None
        if x.order_key is not None:
            # the key (address, version, shape) can be re-used by ANOTHER tensor once the first is freed: a 64-row sample of the prepared
            # coordinates, compared up to the uniform rescaling a new lengthscale applies, says whether it is still the same cloud
            stride = max(1, n // 64)
            samp = x.xp[::stride][:64, : x.d]
            samp = (samp - samp.mean(0, keepdim=True))
            samp = samp / samp.abs().max().clamp_min(1e-30)
            hit = _ORDER_CACHE.get(x.order_key)
            if hit is not None and hit[1].shape == samp.shape and bool(torch.allclose(hit[1], samp, rtol=1e-4, atol=1e-5)):
                perm = hit[0]
        if perm is None:
            perm = hilbert_order(x.xp[:, : x.d])
            if x.order_key is not None:
                while len(_ORDER_CACHE) >= 8:          # a handful of clouds (train / test inputs of a few models), oldest first out
                    _ORDER_CACHE.pop(next(iter(_ORDER_CACHE)))
                _ORDER_CACHE[x.order_key] = (perm, samp)
        xs = x.xp.index_select(0, perm)
        ng, nch = (n + 511) // 512, (n + 127) // 128
        r2_128, r2_grp = self._block_radii(xs, n, dp, ng)
        # class of every 128-row chunk: 0 COMPACT -- it lies in a 512-row group all of whose 128 / 256 / 512-row blocks are within the policy;
        # 1 MEDIUM -- the chunk itself is (against its own centre), its group is not; 2 WIDE.  A compact chunk also qualifies as medium, and
        # anything may be treated as wide.
        grp_ok = (r2_grp <= GRAM_MAX_BLOCK_SQRADIUS).repeat_interleave(4)[:nch]
        cls = torch.where(grp_ok, 0, torch.where(r2_128 <= GRAM_MAX_BLOCK_SQRADIUS, 1, 2))
        top = int(cls.max().item())
        if top > 0:
            # compact groups first (Hilbert order kept), then the medium chunks, then the wide ones.  The compact region must consist of whole
            # 512-row groups and a ragged chunk can only stand at the very end of the order
            if n % 512 != 0:
                cls[4 * (ng - 1):] = cls[4 * (ng - 1):].clamp_min(1)          # a ragged last group: its chunks are medium at best
            if n % 128 != 0:
                cls[-1] = top                                                 # a ragged last chunk joins the last class present
            order = torch.sort(cls, stable=True).indices
            rows = (order.unsqueeze(1) * 128 + torch.arange(128, device=dev).unsqueeze(0)).reshape(-1)
            rows = rows[rows < n]
            perm = perm[rows]
            xs = xs[rows]
            sizes = torch.full((nch,), 128, device=dev, dtype=torch.int64)
            sizes[-1] = n - 128 * (nch - 1)
            n0, n1 = (int(v) for v in torch.stack([sizes[cls == 0].sum(), sizes[cls == 1].sum()]).tolist())
            self.n_compact, self.n_block = n0, n0 + n1
        else:
            self.n_compact = self.n_block = n
        r2c = torch.where(cls == 0, r2_grp.repeat_interleave(4)[:nch], r2_128)[cls < 2]
        self.r2 = float(r2c.max().item()) if self.n_block else 0.0
        self.perm = perm
        self.xs = xs.contiguous()
        ld = round_up(n, 4)
        inv = torch.arange(ld, device=dev, dtype=torch.int64)
        inv[perm] = torch.arange(n, device=dev, dtype=torch.int64)
        self.inv_pad = inv
        nch = (n + 127) // 128
        pad = nch * 128 - n
        xs_pad = torch.cat([self.xs, self.xs[-1:].expand(pad, -1)], 0) if pad else self.xs
        self.centers = xs_pad.reshape(nch, 128, dp).mean(1).contiguous()
        # bounding sphere of every 128-row chunk about its centre (far-pair tile culling, settings.far_pair_cutoff): a hair of slack for the rounding
        # of the norm, so that the sphere CONTAINS the chunk
        self.radii = ((xs_pad.reshape(nch, 128, dp) - self.centers.unsqueeze(1)).pow(2).sum(-1).max(1).values.sqrt() * (1.0 + 1e-6) + 1e-30).contiguous()
        self._perm_pad = None

    @property
    def perm_pad(self):
        """[round_up(n, 4)] gather index that takes a probe-major row in the ORIGINAL order to the sorted order (padding: the last row again)."""
        if self._perm_pad is None:
            ld = self.inv_pad.numel()
            self._perm_pad = torch.cat([self.perm, self.perm[-1:].expand(ld - self.perm.numel())]) if ld > self.perm.numel() else self.perm
        return self._perm_pad


    @staticmethod
    def _block_radii(xs: torch.Tensor, n: int, dp: int, ng: int):
        """(largest |z - centre|^2 of every 128-row chunk of ``xs`` against its own centre [nch];  per 512-row group, the largest over its 128 /
        256 / 512-row blocks [ng]) -- each block against ITS OWN centre as the kernels form it (csrc/gram_f16.hpp ``load_center``): the mean of
        the centres of the block's 128-row chunks, chunk indices clamped to the last chunk; the last chunk is padded with copies of the last row
        (what ``centers`` stores)."""
        dev = xs.device
        nch = (n + 127) // 128
        pad = nch * 128 - n
        xc = (torch.cat([xs, xs[-1:].expand(pad, -1)], 0) if pad else xs).reshape(nch, 128, dp)
        c128 = xc.mean(1)
        per_chunk = []
        for m in (1, 2, 4):
            nb = (nch + m - 1) // m
            idx = (torch.arange(nb, device=dev).unsqueeze(1) * m + torch.arange(m, device=dev).unsqueeze(0)).clamp_max(nch - 1)
            bc = c128[idx].mean(1)                                                        # [nb, dp] block centres
            cb = bc[torch.arange(nch, device=dev) // m]                                    # the centre every chunk is measured against
            per_chunk.append((xc - cb.unsqueeze(1)).pow(2).sum(-1).max(1).values)          # [nch]
        worst = torch.maximum(per_chunk[0], torch.maximum(per_chunk[1], per_chunk[2]))
        return per_chunk[0], torch.nn.functional.pad(worst, (0, ng * 4 - nch)).reshape(ng, 4).max(1).values


REGION_MERGE_MAX_ROWS = 4096   # kv_partials_sorted: medium + wide rows up to this many share ONE direct-difference launch
# kv_partials_sorted: the region launches of a block-centred product (medium / wide rows: a few per cent of the rows on launches that cannot fill
# the chip) go to a SECOND HIP stream and run beside the compact launch instead of behind it -- fork at the start of the product, join before the
# slabs are read (GPAMD_REGION_STREAMS=0: one stream, the round-5 order; off under stream capture and under far-pair culling, whose tile-list
# workspace the launches share)
REGION_STREAMS = os.environ.get("GPAMD_REGION_STREAMS", "1") != "0"
_side_streams: dict = {}


def _side_stream(device) -> "torch.cuda.Stream":
    s = _side_streams.get(device.index)
    if s is None:
        s = _side_streams[device.index] = torch.cuda.Stream(device=device)
    return s
FAR_FEW_MAX_KEPT = 0.3   # far-pair culling: products of fewer than five columns move to the (culled) split kernels below this surviving share of tiles
FAR_MIN_POINTS = 1024   # far-pair culling: smaller clouds are launch-bound, the two extra gathers per product would cost more than any tile saves


def far_sq_cutoff(kind: str, eps: float, param=None) -> float:
    """Squared distance in PREPARED coordinates (csrc/common.hpp: RBF k = 2^-s, Matern k = poly(r) e^-r with r = sqrt(s), RQ k = (1 + s)^-alpha)
    at which the family's covariance falls to ``eps``: beyond it every entry of K is <= eps."""
    if kind == "rbf":
        return math.log2(1.0 / eps)
    if kind == "rq":
        return eps ** (-1.0 / float(param)) - 1.0
    poly = {"matern12": lambda r: 1.0, "matern32": lambda r: 1.0 + r, "matern52": lambda r: 1.0 + r + r * r / 3.0}[kind]
    lo, hi = 0.0, 200.0
    for _ in range(80):                      # k is decreasing in r: bisection
        mid = 0.5 * (lo + hi)
        lo, hi = (mid, hi) if poly(mid) * math.exp(-mid) > eps else (lo, mid)
    return hi * hi


def far_cull(x1: PreparedPoints, x2: PreparedPoints):
    """The squared cutoff of a product k(x1, x2) V under ``settings.far_pair_cutoff``, or None when culling is off or cannot drop anything (small
    clouds; a cloud narrower than the cutoff)."""
    from . import settings

    eps = settings.far_pair_cutoff.value()
    if eps is None or FORCE_KV_FLAGS is not None or not (x1.fused and x2.fused) or min(x1.n, x2.n) < FAR_MIN_POINTS:
        return None
    sq = far_sq_cutoff(x1.kind, float(eps), x1.param)
    z1 = x1.zmax2
    z2 = x2.zmax2 if x2 is not x1 else z1
    if not sq < (math.sqrt(z1) + math.sqrt(z2)) ** 2:
        return None
    return sq


def far_kept_fraction(x1: PreparedPoints, x2: PreparedPoints, sq: float, bm: int = 512) -> float:
    """Share of the (bm-row block, 128-point tile) pairs of k(x1, x2) that survive a squared cutoff ``sq`` -- the test of csrc/kv_cull.hpp
    restated with torch on the sorted views (reporting only: scripts/far_cull_timing.py, the reference-workload record)."""
    sv1, sv2 = x1.sorted_view(), x2.sorted_view()
    m = bm // 128
    nch = sv1.centers.shape[0]
    nb = (nch + m - 1) // m
    idx = (torch.arange(nb, device=sv1.centers.device).unsqueeze(1) * m + torch.arange(m, device=sv1.centers.device).unsqueeze(0)).clamp_max(nch - 1)
    bc = sv1.centers[idx].mean(1)
    br = ((sv1.centers[idx] - bc.unsqueeze(1)).norm(dim=-1) + sv1.radii[idx]).max(1).values
    gap = torch.cdist(bc, sv2.centers) - br.unsqueeze(1) - sv2.radii.unsqueeze(0)
    return float((~((gap > 0) & (gap * gap > sq))).float().mean().item())


def rows_sorted(x1: PreparedPoints, x2: PreparedPoints, flags: int) -> bool:
    """True when :func:`kv_partials_sorted` returns the output rows of k(x1, x2) V in x1's Hilbert order (the caller then un-sorts them)."""
    return far_cull(x1, x2) is not None or gram_operands(x1, x2, flags)[2] is not None


def kind_id(xp: PreparedPoints) -> int:
    """Integer id of the covariance family of ``xp`` for the C ABI."""
    return KIND_IDS[xp.kind]


def kind_args(xp: PreparedPoints):
    """(kind, kparam): the two leading arguments of every float32 entry point that evaluates the covariance (ABI version 2: the shape
    parameter of a parametrised family -- RQ's alpha -- is an explicit argument, the library keeps no per-thread kernel state)."""
    return KIND_IDS[xp.kind], float(xp.param) if xp.param is not None else 0.0


def _split_on() -> bool:
    """The split-operand contraction (f16 matrix pipe at f32 accuracy) is selected: ``settings.split_contraction`` unless a test forces it."""
    if SPLIT_CONTRACTION is not None:
        return bool(SPLIT_CONTRACTION)
    from . import settings

    return settings.split_contraction.on()


def gram_mode(x1: PreparedPoints, x2: PreparedPoints) -> int:
    """How the Gram-form kernels may evaluate the squared distances of k(x1, x2): 1 = cloud-centred quadratic expansion (both clouds
    within max |z|^2 <= 32 of the common origin), 2 = BLOCK-centred expansion on the Hilbert-sorted rows of x1 (any cloud width: the
    error scales with the radius of a 128..512-row block of x1, the reference's Gram-trick distance has no scale limit either,
    ``gpytorch/kernels/kernel.py:26-49``), 0 = neither (direct-difference kernels; a warning is issued once per reason)."""
    z1 = x1.zmax2
    z2 = x2.zmax2 if x2 is not x1 else z1
    if max(z1, z2) <= GRAM_MAX_SQNORM:
        return 1
    if (math.sqrt(z1) + math.sqrt(z2)) ** 2 <= (GRAM_MAX_EXTENT_SQ_RQ if x1.kind == "rq" else GRAM_MAX_EXTENT_SQ) and x1.n >= 128:
        sv = x1.sorted_view()
        if x1.n - sv.n_block <= GRAM_MAX_WIDE_FRACTION * x1.n:
            return 2
        reason = (f"{x1.n - sv.n_block} of {x1.n} points (d = {x1.d}) lie in 128-point runs of the Hilbert order wider than "
                  f"|z - centre|^2 = {GRAM_MAX_BLOCK_SQRADIUS}")
    else:
        reason = f"cloud extent (max |x / lengthscale|^2 = {max(z1, z2):.0f}) outside the range of the split-f16 expansion"
    key = (x1.kind, x1.d, reason[:24])
    if key not in _warned_fallback:
        import warnings

        _warned_fallback.add(key)
        warnings.warn("gpytorch_amd: kernel products fall back to the direct-difference kernels (squared distances on the vector pipe: slower than "
                      f"the Gram-form kernels, 1.1-1.6x per row with the split contraction, 2-3x without): {reason}", RuntimeWarning)
    return 0


def kv_flags(x1: PreparedPoints, x2: PreparedPoints, t: int) -> int:
    """Select the Gram-form generation kernel when it is both applicable and accurate (see kv_gram.hpp)."""
    if not (x1.fused and x2.fused):
        return 0
    if FORCE_KV_FLAGS is not None:
        return FORCE_KV_FLAGS
    if SPLIT_CONTRACTION is None:
        # no size rule: the three extra launches of the pre-pass do not make small products slower -- measured, mBCG per iteration
        # with the fp32-MFMA contraction instead: 107 vs 57 us at n = 2000, 231 vs 105 us at n = 5000, 229 vs 144 us at n = 20 000
        # (eleven columns; profiles/r02_s33_cg_small_n_fp32_contraction.json vs r02_s30_*)
        split = _split_on()
    else:
        split = SPLIT_CONTRACTION
    # far-pair culling in force for this product (settings.far_pair_cutoff): products of fewer than five columns go to the split kernels too -- where
    # few enough tiles survive: a mostly empty 32-column tile costs 2.5-3x a few-column kernel per visited tile (road3d-shaped cloud: the culled
    # one-column product wins below ~0.35 surviving, and is 1.3x SLOWER at lengthscale 0.35 where 0.98 survive)
    few = 0
    if split and t < 5:
        sq = far_cull(x1, x2)
        if sq is not None:
            key = (id(x2), round(sq, 6))
            if key not in x1._far_keep:
                x1._far_keep[key] = far_kept_fraction(x1, x2, sq, 512)
            few = KV_SPLIT_FEW if x1._far_keep[key] < FAR_FEW_MAX_KEPT else 0
    if x1.kind == "matern12" or gram_mode(x1, x2) == 0:
        # direct differences: the contraction still goes to the f16 matrix pipe (csrc/kv_directh.hpp, 5 .. 65 columns per group, d <= 10)
        return (KV_SPLIT | few) if (split and x1.d <= DIRECT_SPLIT_MAX_DIM) else 0
    return KV_GRAM | ((KV_SPLIT | few) if split else 0)


def prep_points(kind: str, x: torch.Tensor, lengthscale: torch.Tensor, shift: torch.Tensor | None = None, param=None) -> PreparedPoints:
    """x: [n, d]; lengthscale: 1 or d values (any shape); shift: d values or None; param: shape parameter (RQ: alpha)."""
    _require_gpu(x, "x")
    if kind == "rq":
        if param is None:
            raise ValueError("the rational-quadratic family needs its shape parameter alpha")
        param = float(param)
        if not param > 0.0:
            raise ValueError("the rational-quadratic shape parameter alpha must be positive")
    else:
        param = None
    n, d = x.shape[-2], x.shape[-1]
    dp = padded_dim(d)
    wd = work_dtype(x)
    x_src = x if x.dim() == 2 else None
    x = x.detach().to(wd).contiguous()
    ls = lengthscale.detach().to(device=x.device, dtype=wd).reshape(-1).contiguous()
    if ls.numel() not in (1, d):
        raise ValueError(f"lengthscale must have 1 or {d} elements, got {ls.numel()}")
    sh = None if shift is None else shift.detach().to(device=x.device, dtype=wd).reshape(-1).contiguous()
    xp = torch.empty(n, dp, device=x.device, dtype=wd)
    fn = lib().gpamd_prep_points_f64 if wd == torch.float64 else lib().gpamd_prep_points_f32
    lead = (KIND_IDS[kind], param if param is not None else 0.0)
    check(fn(*lead, _ptr(x), n, d, x.stride(0), _ptr(ls), ls.numel(), _ptr(sh), _ptr(xp), dp, _stream(x.device)), "prep_points")
    out = PreparedPoints(xp, n, d, dp, kind, param)
    if ls.numel() == 1 and x_src is not None:
        out.order_key = (x_src.data_ptr(), x_src._version, tuple(x_src.shape), str(x_src.device), x_src.dtype)
    return out


def to_probe_major(rhs: torch.Tensor, dtype: torch.dtype | None = None) -> torch.Tensor:
    """[n, t] (any strides) -> [t, ld] with ld = round_up(n, 4), zero padded; float32 unless ``dtype`` says otherwise
    (callers on the generic path pass the dtype of their prepared points)."""
    n, t = rhs.shape[-2], rhs.shape[-1]
    ld = round_up(n, 4)
    out = torch.zeros(t, ld, device=rhs.device, dtype=torch.float32 if dtype is None else dtype)
    out[:, :n] = rhs.detach().t()
    return out


def from_probe_major(vt: torch.Tensor, n: int) -> torch.Tensor:
    """[t, ld] -> [n, t] contiguous."""
    return vt[:, :n].t().contiguous()


_ws_cache: dict = {}


def workspace(device, nfloats: int, slot: int = 0) -> torch.Tensor:
    """A cached float32 scratch buffer per (device, slot): slot 0 holds the partial slabs of a product, slot 1 the wide-row launch's own slabs."""
    key = (device.type, device.index, slot)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nfloats:
        if buf is not None and device.type == "cuda":
            torch.cuda.synchronize(device)   # the buffer being dropped may still be read on another stream than the allocating one (region streams)
        buf = torch.empty(max(nfloats, 1 << 20), device=device, dtype=torch.float32)
        _ws_cache[key] = buf
    return buf


_PLAN_CACHE: dict = {}


def kv_plan(kind: str, n: int, m: int, d: int, t: int, flags: int, ldo: int):
    """(S, jchunk, workspace floats) for the kernel variant that (kind, d, t, flags) selects.  A pure function of its arguments on a given
    device (``gpamd_kv_plan`` asks the runtime for the kernel's occupancy): cached -- the region launches of a block-centred product
    (``_kv_region``) used to repeat the query on every mBCG iteration (advisor finding, round 5)."""
    key = (kind, n, m, d, t, flags, ldo, torch.cuda.current_device() if torch.cuda.is_available() else -1)
    hit = _PLAN_CACHE.get(key)
    if hit is None:
        S, jc, ws = C.c_int(0), C.c_int(0), C.c_int64(0)
        check(lib().gpamd_kv_plan(KIND_IDS[kind], n, m, d, t, flags, ldo, C.byref(S), C.byref(jc), C.byref(ws)), "kv_plan")
        if len(_PLAN_CACHE) > 4096:
            _PLAN_CACHE.clear()
        hit = _PLAN_CACHE[key] = (S.value, jc.value, ws.value)
    return hit


def gram_operands(x1: PreparedPoints, x2: PreparedPoints, flags: int):
    """(X1 array, chunk centres or None, un-sort index or None, rows of the compact region, rows of the compact + medium regions) for a
    Gram-form launch of k(x1, x2).  Block-centred mode: the OUTPUT rows come out in x1's sorted order -- ``unsort`` (an index for
    ``index_select(1, .)`` on a probe-major [t, round_up(n, 4)] block) takes them back; x2 and the right-hand sides stay in their original order."""
    if (flags & KV_GRAM) and FORCE_KV_FLAGS is None and gram_mode(x1, x2) == 2:
        sv = x1.sorted_view()
        return sv.xs, sv.centers, sv.inv_pad, sv.n_compact, sv.n_block
    return x1.xp, None, None, x1.n, x1.n


_far_ws: dict = {}   # per device: the tile-list workspace of the far-pair launches (stream-ordered re-use, like the partial slabs)


def _kv_launch(x1, x2, X1ptr, n_r: int, X2, Xcptr, vt, t: int, Pptr, ldo: int, S: int, jc: int, flags: int, done_ptr, st, cull, row0: int, what: str):
    """One ``gpamd_kv_partials_f32`` launch group; with ``cull`` = (sq, sv1, sv2) the far-pair variant with the bounding spheres of the rows from
    ``row0`` (a multiple of 128) on."""
    L = lib()
    if cull is None:
        check(L.gpamd_kv_partials_f32(*kind_args(x1), X1ptr, n_r, _ptr(X2), x2.n, x1.d, Xcptr, _ptr(vt), vt.stride(0), t, Pptr, ldo, S, jc, flags,
                                      done_ptr, st), what)
        return
    sq, sv1, sv2 = cull
    assert row0 % 128 == 0
    rc = C.c_void_p(sv1.centers.data_ptr() + 4 * x1.dp * (row0 // 128))
    rr = C.c_void_p(sv1.radii.data_ptr() + 4 * (row0 // 128))
    tws = _far_tile_ws(vt.device, int(L.gpamd_kv_far_workspace_ints(n_r, S, jc)))
    check(L.gpamd_kv_partials_far_f32(*kind_args(x1), X1ptr, n_r, _ptr(X2), x2.n, x1.d, Xcptr, _ptr(vt), vt.stride(0), t, Pptr, ldo, S, jc, flags,
                                      done_ptr, st, rc, rr, _ptr(sv2.centers), _ptr(sv2.radii), float(sq), _ptr(tws), tws.numel()), what)


def _kv_region(x1, x2, X1, Xc, row0: int, n_r: int, flags_r: int, vt, t: int, P, ldo: int, S: int, jc: int, done_ptr, st, slot: int, what: str,
               X2=None, cull=None):
    """Rows [row0, row0 + n_r) of a block-centred product (row0 a multiple of 512) on the kernels ``flags_r`` selects, into the shared slabs P.
    A region is a SMALLER product than the compact launch, possibly on another kernel: with that launch's split count it would leave most of the
    chip idle -- measured on the reference's own workloads (profiles/r05_s1_workload_*_kernel_stats.csv): 11.3 ms for 10 % of the rows against
    11.8 ms for the other 90 % (road3d shape), 0.83 against 0.31 ms (protein shape).  So it gets its own plan and, unless the plan coincides, its
    own slabs (workspace ``slot``; always with the split-operand planes, which live behind the slabs): their sum lands in slab 0 of the shared
    layout, the other slabs' rows are zeroed."""
    L = lib()
    assert row0 % 4 == 0
    X2 = x2.xp if X2 is None else X2
    x1r = C.c_void_p(X1.data_ptr() + 4 * x1.dp * row0)
    xcr = C.c_void_p(Xc.data_ptr() + 4 * x1.dp * (row0 // 128)) if (Xc is not None and (flags_r & KV_GRAM)) else None
    ldr = round_up(n_r, 4)
    Sr, jcr, wsr = kv_plan(x1.kind, n_r, x2.n, x1.d, t, flags_r, ldr)
    if Sr == S and jcr == jc and not (flags_r & KV_SPLIT):
        _kv_launch(x1, x2, x1r, n_r, X2, xcr, vt, t, C.c_void_p(P.data_ptr() + 4 * row0), ldo, S, jc, flags_r, done_ptr, st, cull, row0,
                   f"kv_partials ({what})")
        return
    Pr = workspace(vt.device, wsr, slot=slot)
    _kv_launch(x1, x2, x1r, n_r, X2, xcr, vt, t, _ptr(Pr), ldr, Sr, jcr, flags_r, done_ptr, st, cull, row0, f"kv_partials ({what})")
    check(L.gpamd_kv_reduce_f32(_ptr(Pr), Sr, ldr, t, n_r, None, None, None, None, 0, C.c_void_p(P.data_ptr() + 4 * row0), ldo, done_ptr, st),
          f"kv_reduce ({what})")
    if S > 1:
        P[: S * t * ldo].view(S, t, ldo)[1:, :, row0 : row0 + n_r].zero_()


def kv_partials_sorted(x1: PreparedPoints, x2: PreparedPoints, vt: torch.Tensor, t: int, flags: int, P, ldo: int, S: int, jc: int, done_ptr, st):
    """One fused K*V launch group into the partial slabs P; returns the un-sort index of the output rows (None: original order).
    Block-centred mode (``SortedView``): the compact rows on the Gram-form kernels; the rows of the MEDIUM groups (128-row blocks within the
    policy, larger ones not) on the kernels that centre 128-row blocks (flag GPAMD_KV_BLOCK128: the split kernel at one row tile per wave; column
    groups it does not serve fall to the direct-difference kernels inside the library); the rows of the WIDE groups on the direct-difference
    kernels -- further launches into the same slabs."""
    X1, Xc, unsort, n_c, n_b = gram_operands(x1, x2, flags)
    if n_c < n_b < x1.n and x1.n - n_c <= REGION_MERGE_MAX_ROWS:
        # a few medium AND a few wide rows (protein-shaped cloud: 2048 + 232 of 36 584): one direct-difference launch for both instead of two region
        # launches -- each region costs five launches per product (split pre-pass, kernel, slab reduction, zeroing), and at this size the products are
        # launch-bound; direct differences are valid for any row (and 1.1-1.6x slower per row: nothing for a few thousand rows)
        n_b = n_c
    X2, cull = x2.xp, None
    sq = far_cull(x1, x2)
    if sq is not None:
        # far-pair tile culling (settings.far_pair_cutoff): BOTH clouds in Hilbert order -- the rows as in block-centred mode (whatever the
        # generation mode: the cloud-centred expansion and the direct differences accept any row order), the contracted cloud and the columns of V
        # through x2's own sorted view (one gather of t x m floats per product)
        sv1, sv2 = x1.sorted_view(), x2.sorted_view()
        if unsort is None:
            X1, unsort = sv1.xs, sv1.inv_pad
        X2 = sv2.xs
        vt = vt.index_select(1, sv2.perm_pad)
        cull = (sq, sv1, sv2)
    def regions(st_r):
        if n_b > n_c:
            _kv_region(x1, x2, X1, Xc, n_c, n_b - n_c, flags | KV_BLOCK128, vt, t, P, ldo, S, jc, done_ptr, st_r, 1, "medium rows", X2, cull)
        if n_b < x1.n:
            wflags = (flags & (KV_SPLIT | KV_SPLIT_FEW)) if ((flags & KV_SPLIT) and x1.d <= DIRECT_SPLIT_MAX_DIM) else 0   # direct differences (+ the split contraction, kv_directh.hpp)
            _kv_region(x1, x2, X1, None, n_b, x1.n - n_b, wflags, vt, t, P, ldo, S, jc, done_ptr, st_r, 2, "wide rows", X2, cull)

    main = torch.cuda.current_stream(vt.device)
    if (REGION_STREAMS and n_c and n_c < x1.n and cull is None and st is not None and (st.value or 0) == main.cuda_stream
            and not torch.cuda.is_current_stream_capturing()):
        # fork: the region launches (own workspaces, their own rows of the shared slabs) on the side stream, issued FIRST so that their short
        # workgroups take their slots while the compact launch's pre-pass runs; join before anything reads the slabs.  In-order side stream:
        # the region workspaces of consecutive products never overlap; P's region rows were last read by the previous product's reduction,
        # which precedes the fork on the main stream.
        side = _side_stream(vt.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            regions(C.c_void_p(side.cuda_stream))
        _kv_launch(x1, x2, _ptr(X1), n_c, X2, _ptr(Xc), vt, t, _ptr(P), ldo, S, jc, flags, done_ptr, st, cull, 0, "kv_partials")
        main.wait_stream(side)
        return unsort
    if n_c:
        _kv_launch(x1, x2, _ptr(X1), n_c, X2, _ptr(Xc), vt, t, _ptr(P), ldo, S, jc, flags, done_ptr, st, cull, 0, "kv_partials")
    regions(st)
    return unsort


def kv(x1: PreparedPoints, x2: PreparedPoints, vt: torch.Tensor, scale=None, dscale=None, vd=None, out=None, dvec=None):
    """out[t, ld_n] = scale * k(x1, x2) @ V + (dscale + dvec) .* Vd in probe-major layout.

    vt: [t, ldv] with ldv >= m; scale/dscale: 1-element device tensors or None; dvec: float32 [>= n] or None."""
    _require_gpu(vt, "vt")
    assert x1.kind == x2.kind and x1.dp == x2.dp and x1.dtype == x2.dtype
    wd = x1.dtype
    vd_is_vt = vd is vt
    vt = vt if vt.dtype == wd else vt.to(wd)
    vd = vt if vd_is_vt else (vd if vd is None or vd.dtype == wd else vd.to(wd))
    scale, dscale, dvec = (a if a is None or a.dtype == wd else a.to(wd) for a in (scale, dscale, dvec))
    if not (x1.fused and x2.fused):
        return kv_generic(x1, x2, vt, scale, dscale, vd, out, dvec)
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
    unsort = kv_partials_sorted(x1, x2, vt, t, flags, ws, ldo, S, jc, None, st)
    if unsort is not None:
        # block-centred Gram expansion: the slabs hold the rows in x1's Hilbert order -> sum them, take the rows back (one gather of
        # t x n floats against n m t pair evaluations), then apply the diagonal epilogue in the original order
        tmp = torch.empty(t, ldo, device=vt.device, dtype=torch.float32)
        check(L.gpamd_kv_reduce_f32(_ptr(ws), S, ldo, t, n, _ptr(scale), None, None, None, 0, _ptr(tmp), ldo, None, st), "kv_reduce")
        res = tmp.index_select(1, unsort)
        if vd is not None and (dscale is not None or dvec is not None):
            dtot = torch.zeros(n, device=vt.device, dtype=res.dtype)
            if dscale is not None:
                dtot += dscale.reshape(())
            if dvec is not None:
                dtot += dvec[:n]
            res[:, :n].addcmul_(vd[:, :n], dtot)
        if out.shape == res.shape and out.stride(1) == 1:
            out.copy_(res)
            return out
        return res
    check(
        L.gpamd_kv_reduce_f32(
            _ptr(ws), S, ldo, t, n, _ptr(scale), _ptr(dscale), _ptr(dvec), _ptr(vd), 0 if vd is None else vd.stride(0),
            _ptr(out), out.stride(0), None, st,
        ),
        "kv_reduce",
    )
    return out


def kernel_row_block(x1: PreparedPoints, r0: int, nrows: int, x2: PreparedPoints, scale=None) -> torch.Tensor:
    """Dense rows [r0, r0 + nrows) of scale * k(x1, x2) (HIP generation kernels, float32 or float64)."""
    out = torch.empty(nrows, x2.n, device=x1.xp.device, dtype=x1.dtype)
    st = _stream(out.device)
    if x1.dtype == torch.float64:
        sc = None if scale is None else scale.to(torch.float64)
        check(lib().gpamd_kernel_rows_f64(*kind_args(x1), _ptr(x1.xp), None, r0, nrows, _ptr(x2.xp), x2.n, x1.dp, _ptr(sc),
                                          _ptr(out), out.stride(0), st), "kernel_rows_f64")
    else:
        blk = x1.xp[r0 : r0 + nrows]
        check(lib().gpamd_kernel_dense_f32(*kind_args(x1), _ptr(blk), nrows, _ptr(x2.xp), x2.n, x1.dp, _ptr(scale), _ptr(out),
                                           out.stride(0), st), "kernel_dense")
    return out


FUSED_F64_MAX_DP = 16   # kv_f64.hpp keeps NI * dp doubles of x_i in registers (two row tiles per wave above dp = 8)
FORCE_CHUNKED = False  # tests: keep float64 products on the row-block path


def fused_f64(x1: PreparedPoints, x2: PreparedPoints) -> bool:
    """float64 clouds with d <= 16: fused generation + float64 MFMA contraction (csrc/kv_f64.hpp)."""
    return (x1.dtype == torch.float64 and x2.dtype == torch.float64 and x1.dp == x2.dp and x1.dp <= FUSED_F64_MAX_DP
            and not FORCE_CHUNKED)


def kv_partials_f64(x1: PreparedPoints, x2: PreparedPoints, vt: torch.Tensor, done_ptr=None):
    """Unscaled partial slabs of k(x1, x2) @ V from the fused float64 kernel: (P [S*t*ldo doubles], S, ldo)."""
    n, m, t = x1.n, x2.n, vt.shape[0]
    ldo = round_up(n, 4)
    S, jc, ws = C.c_int(), C.c_int(), C.c_int64()
    check(lib().gpamd_kv_plan_f64(n, m, x1.dp, t, ldo, C.byref(S), C.byref(jc), C.byref(ws)), "kv_plan_f64")
    key = ("f64", vt.device)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < ws.value:
        buf = torch.empty(max(ws.value, 1 << 18), device=vt.device, dtype=torch.float64)
        _ws_cache[key] = buf
    check(lib().gpamd_kv_partials_f64(*kind_args(x1), _ptr(x1.xp), n, _ptr(x2.xp), m, x1.dp, _ptr(vt), vt.stride(0), t, _ptr(buf),
                                      ldo, S.value, jc.value, done_ptr, _stream(vt.device)), "kv_partials_f64")
    return buf, S.value, ldo


def kv_generic(x1: PreparedPoints, x2: PreparedPoints, vt: torch.Tensor, scale=None, dscale=None, vd=None, out=None, dvec=None):
    """Products outside the fused float32 kernels: fused float64 (d <= 8) or row blocks x GEMM (kv_chunked)."""
    if not fused_f64(x1, x2):
        return kv_chunked(x1, x2, vt, scale, dscale, vd, out, dvec)
    dt = torch.float64
    vt = vt if vt.dtype == dt else vt.to(dt)
    n, t = x1.n, vt.shape[0]
    P, S, ldo = kv_partials_f64(x1, x2, vt)
    if out is None:
        out = torch.empty(t, ldo, device=vt.device, dtype=dt)
    sc, ds, dv = (None if a is None else a.to(dt) for a in (scale, dscale, dvec))
    vdd = None if vd is None else (vt if vd is vt else vd.to(dt))
    check(lib().gpamd_kv_reduce_f64(_ptr(P), S, ldo, t, n, _ptr(sc), _ptr(ds), _ptr(dv), _ptr(vdd), 0 if vdd is None else vdd.stride(0),
                                    _ptr(out), out.stride(0), None, _stream(vt.device)), "kv_reduce_f64")
    return out


def kv_chunked(x1: PreparedPoints, x2: PreparedPoints, vt: torch.Tensor, scale=None, dscale=None, vd=None, out=None, dvec=None):
    """Generic-path product (float64, or d > 16): K is generated in dense row blocks by the HIP kernels and
    multiplied with a library GEMM -- the reference's chunked strategy (lazy_evaluated_kernel_tensor.py:245-275)
    on the device.  HBM-bound: itemsize * n * m bytes written and read per product."""
    n, m = x1.n, x2.n
    t = vt.shape[0]
    dt = x1.dtype
    ldo = round_up(n, 4)
    if out is None:
        out = torch.zeros(t, ldo, device=vt.device, dtype=dt)
    v = vt[:, :m].to(dt)
    rows = int(max(1, min(n, 65535, (1 << 27) // max(m, 1))))
    for r0 in range(0, n, rows):
        nr = min(rows, n - r0)
        kc = kernel_row_block(x1, r0, nr, x2)
        torch.matmul(v, kc.t(), out=out[:, r0 : r0 + nr])
    if scale is not None:
        out[:, :n].mul_(scale.to(dt).reshape(()))
    if vd is not None:
        coef = torch.zeros((), device=out.device, dtype=dt) if dscale is None else dscale.to(dt).reshape(())
        if dvec is not None:
            out[:, :n].add_((coef + dvec[:n].to(dt)) * vd[:, :n].to(dt))
        else:
            out[:, :n].add_(coef * vd[:, :n].to(dt))
    return out


def kernel_dense(x1: PreparedPoints, x2: PreparedPoints, scale=None) -> torch.Tensor:
    if x1.dtype == torch.float64:
        outs = [kernel_row_block(x1, r0, min(65535, x1.n - r0), x2, scale) for r0 in range(0, x1.n, 65535)]
        return outs[0] if len(outs) == 1 else torch.cat(outs, 0)
    out = torch.empty(x1.n, x2.n, device=x1.xp.device, dtype=torch.float32)
    check(
        lib().gpamd_kernel_dense_f32(
            *kind_args(x1), _ptr(x1.xp), x1.n, _ptr(x2.xp), x2.n, x1.dp, _ptr(scale), _ptr(out), out.stride(0),
            _stream(out.device),
        ),
        "kernel_dense",
    )
    return out


def kernel_rows(x1: PreparedPoints, rows: torch.Tensor, x2: PreparedPoints, scale=None) -> torch.Tensor:
    rows = rows.to(device=x1.xp.device, dtype=torch.int64).contiguous()
    if x1.dtype == torch.float64:
        out = torch.empty(rows.numel(), x2.n, device=x1.xp.device, dtype=torch.float64)
        sc = None if scale is None else scale.to(torch.float64)
        check(lib().gpamd_kernel_rows_f64(*kind_args(x1), _ptr(x1.xp), _ptr(rows), 0, rows.numel(), _ptr(x2.xp), x2.n, x1.dp,
                                          _ptr(sc), _ptr(out), out.stride(0), _stream(out.device)), "kernel_rows_f64")
        return out
    out = torch.empty(rows.numel(), x2.n, device=x1.xp.device, dtype=torch.float32)
    check(
        lib().gpamd_kernel_rows_f32(
            *kind_args(x1), _ptr(x1.xp), _ptr(rows), rows.numel(), _ptr(x2.xp), x2.n, x1.dp, _ptr(scale), _ptr(out),
            out.stride(0), _stream(out.device),
        ),
        "kernel_rows",
    )
    return out


def kernel_diag(x1: PreparedPoints, x2: PreparedPoints, scale=None) -> torch.Tensor:
    assert x1.n == x2.n
    if x1.dtype == torch.float64:
        out = torch.empty(x1.n, device=x1.xp.device, dtype=torch.float64)
        sc = None if scale is None else scale.to(torch.float64)
        check(lib().gpamd_kernel_diag_f64(*kind_args(x1), _ptr(x1.xp), _ptr(x2.xp), x1.n, x1.dp, _ptr(sc), _ptr(out),
                                          _stream(out.device)), "kernel_diag_f64")
        return out
    out = torch.empty(x1.n, device=x1.xp.device, dtype=torch.float32)
    check(
        lib().gpamd_kernel_diag_f32(
            *kind_args(x1), _ptr(x1.xp), _ptr(x2.xp), x1.n, x1.dp, _ptr(scale), _ptr(out), _stream(out.device)
        ),
        "kernel_diag",
    )
    return out


def coldot(a: torch.Tensor, b: torch.Tensor, n: int) -> torch.Tensor:
    """Per-row (probe-major) inner products over the first n entries: out[c] = <a[c,:n], b[c,:n]>."""
    t = a.shape[0]
    if a.dtype != b.dtype:
        b = b.to(a.dtype)
    out = torch.empty(t, device=a.device, dtype=a.dtype)
    scratch = torch.empty(t * 256, device=a.device, dtype=a.dtype)
    fn = lib().gpamd_coldot_f64 if a.dtype == torch.float64 else lib().gpamd_coldot_f32
    check(fn(_ptr(a), _ptr(b), a.stride(0), n, t, _ptr(out), _ptr(scratch), _stream(a.device)), "coldot")
    return out


def pivoted_cholesky(xp: PreparedPoints, scale, rank: int, tol: float):
    """Returns (L [m, n] row-major view of the rank-m factor (L^T of the reference's n x m), pivots[m], m)."""
    n = xp.n
    rank = min(rank, n)
    dev = xp.xp.device
    if xp.dtype != torch.float32:
        # float64 models: the greedy factor is built from a float32 copy of the prepared points.  Any SPD
        # P = L L^T + s2 I is a valid preconditioner; everything derived from this L (Q1, log|P|, the probe
        # covariance) is then computed in float64 by the caller, so the float64 solve loses nothing.
        xp32 = PreparedPoints(xp.xp.to(torch.float32), n, xp.d, xp.dp, xp.kind, xp.param)
        L32, piv, m = pivoted_cholesky(xp32, None if scale is None else scale.to(torch.float32), rank, tol)
        return L32.to(xp.dtype), piv, m
    ldl = round_up(n, 4)
    L = torch.zeros(rank, ldl, device=dev, dtype=torch.float32)
    piv = torch.zeros(rank, device=dev, dtype=torch.int64)
    fwork = torch.empty(n + 4, device=dev, dtype=torch.float32)
    iwork = torch.zeros(2 + 2 * n, device=dev, dtype=torch.int32)
    check(
        lib().gpamd_pivoted_cholesky_f32(
            *kind_args(xp), _ptr(xp.xp), n, xp.dp, _ptr(scale), rank, float(tol), _ptr(L), ldl, _ptr(piv), _ptr(fwork),
            _ptr(iwork), _stream(dev),
        ),
        "pivoted_cholesky",
    )
    m = int(iwork[0].item())
    return L[:m, :n], piv[:m], m


def _far_tile_ws(dev, nints: int) -> torch.Tensor:
    tws = _far_ws.get(dev)
    if tws is None or tws.numel() < nints:
        tws = _far_ws[dev] = torch.empty(max(nints, 1 << 18), device=dev, dtype=torch.int32)
    return tws


def far_sorted_right(x2: PreparedPoints, rt: torch.Tensor):
    """(sorted x2 array, rt with its columns in x2's Hilbert order, zero-padded to a multiple of four, x2's sorted view) for a culled derivative call."""
    sv2 = x2.sorted_view()
    rts = rt[:, : x2.n].index_select(1, sv2.perm)
    if rts.shape[1] % 4:
        rts = torch.nn.functional.pad(rts, (0, 4 - rts.shape[1] % 4))
    return sv2.xs, rts.contiguous(), sv2


def kv_grad(x1: PreparedPoints, x2: PreparedPoints, lt: torch.Tensor, rt: torch.Tensor, iso: bool = False, far=None) -> torch.Tensor:
    """Fused bilinear derivative with W = lt^T rt (lt: [t, ld_n] over x1, rt: [t, ld_m] over x2).

    Returns float32 [1 + dp]:  g[0] = sum_ij W_ij k_ij;  g[1+q] = sum_ij W_ij dk/ds_ij (z_iq - z_jq)^2
    where z are the PREPARED coordinates and s the squared prepared distance.

    Far-pair culling (``settings.far_pair_cutoff``): both clouds and their vector blocks go to the kernel in Hilbert order (the sums are order-free) with the
    bounding spheres of their 128-point chunks.  ``far`` = (sq, row_centres, row_radii, X2 sorted, rt sorted, sv2): the caller (``kv_grad2``, for its wide
    rows) has x1 / lt in curve order already and hands over the spheres of those rows and the sorted right operands."""
    _require_gpu(lt, "left")
    assert x1.kind == x2.kind and x1.dp == x2.dp and lt.shape[0] == rt.shape[0] and x1.fused and x2.fused
    lt = lt if lt.dtype == torch.float32 else lt.to(torch.float32)
    rt = rt if rt.dtype == torch.float32 else rt.to(torch.float32)
    t = lt.shape[0]
    dev = lt.device
    X1, X2 = x1.xp, x2.xp
    cull = None
    if far is not None:
        sq, rc, rr, X2, rt, sv2 = far
        cull = (sq, rc, rr, sv2)
    else:
        sq = far_cull(x1, x2)
        if sq is not None:
            sv1 = x1.sorted_view()
            X1 = sv1.xs
            lt = lt[:, : x1.n].index_select(1, sv1.perm)
            if lt.shape[1] % 4:
                lt = torch.nn.functional.pad(lt, (0, 4 - lt.shape[1] % 4))
            lt = lt.contiguous()
            X2, rt, sv2 = far_sorted_right(x2, rt)
            cull = (sq, sv1.centers, sv1.radii, sv2)
    nd = int(lib().gpamd_kv_grad_workspace_doubles(x1.n, x2.n, t, x1.dp))
    ws = torch.empty(nd, device=dev, dtype=torch.float64)
    out = torch.empty(1 + x1.dp, device=dev, dtype=torch.float32)
    if cull is None:
        check(
            lib().gpamd_kv_grad_f32(
                kind_id(x1), _ptr(X1), x1.n, _ptr(X2), x2.n, x1.dp, _ptr(lt), lt.stride(0), _ptr(rt), rt.stride(0),
                t, 1 if iso else 0, _ptr(out), _ptr(ws), nd, _stream(dev),
            ),
            "kv_grad",
        )
    else:
        sq, rc, rr, sv2 = cull
        tws = _far_tile_ws(dev, int(lib().gpamd_kv_grad_far_workspace_ints(x1.n, x2.n)))
        check(
            lib().gpamd_kv_grad_far_f32(
                kind_id(x1), _ptr(X1), x1.n, _ptr(X2), x2.n, x1.dp, _ptr(lt), lt.stride(0), _ptr(rt), rt.stride(0),
                t, 1 if iso else 0, _ptr(out), _ptr(ws), nd, _stream(dev), _ptr(rc), _ptr(rr), _ptr(sv2.centers), _ptr(sv2.radii), float(sq),
                _ptr(tws), tws.numel(),
            ),
            "kv_grad (far-pair culling)",
        )
    return out


FORCE_GRAD_DIRECT = False  # tests: keep the bilinear derivative on the direct-difference kernel (kv_grad.hpp)
GRAD_SPLIT_MIN_COLS = 24   # backward: columns from which W = L^T R runs on hi/lo-split f16 operands (kv_grad2 below)
GRAD_SPLIT_MAX_ARD_DIM = 16  # ... in the per-dimension mode (ARD / input gradients) up to this many dimensions (tuning knob: 6 sends d > 6 back to the fp32 W contraction)


def grad_gram_ok(x1: PreparedPoints, x2: PreparedPoints) -> bool:
    """The Gram-form derivative kernel (kv_grad2.hpp) applies: fused float32 clouds, not Matern-1/2, cloud- or block-centred expansion
    within its accuracy policy (``gram_mode``)."""
    if not (x1.fused and x2.fused) or x1.kind == "matern12" or (FORCE_GRAD_DIRECT and x1.kind != "rq"):
        return False
    return gram_mode(x1, x2) != 0


def kv_grad2(x1: PreparedPoints, x2: PreparedPoints, lt: torch.Tensor, rt: torch.Tensor, iso: bool = False, want_gz1: bool = False):
    """Gram-form fused bilinear derivative (same return convention as :func:`kv_grad`), optionally with the gradient with
    respect to the PREPARED left points: returns (g float32 [1 + dp], gz1 float32 [n, d] or None)."""
    _require_gpu(lt, "left")
    assert x1.kind == x2.kind and x1.dp == x2.dp and lt.shape[0] == rt.shape[0]
    lt = lt if lt.dtype == torch.float32 else lt.to(torch.float32)
    rt = rt if rt.dtype == torch.float32 else rt.to(torch.float32)
    t, dev, L = lt.shape[0], lt.device, lib()
    nd = int(L.gpamd_kv_grad2_workspace_doubles(x1.n, x2.n, t, x1.d))
    ws = torch.empty(nd, device=dev, dtype=torch.float64)
    out = torch.empty(2 + x1.dp, device=dev, dtype=torch.float32)   # [1 + dp]: shape-parameter sum (RQ)
    gzt = xws = None
    nx = 0
    ldg = round_up(x1.n, 4)
    if want_gz1:
        nx = int(L.gpamd_kv_grad2_xworkspace_floats(x1.n, x2.n, t, x1.d))
        xws = torch.empty(nx, device=dev, dtype=torch.float32)
        gzt = torch.empty(x1.d, ldg, device=dev, dtype=torch.float32)
    if gram_mode(x1, x2) == 0 and max(x1.zmax2, x2.zmax2) > GRAM_MAX_SQNORM:
        # (callers check grad_gram_ok first; a direct call outside the accuracy policy of the quadratic expansion must not pass silently)
        raise RuntimeError("kv_grad2: the clouds are outside the accuracy policy of the Gram-form expansion (backend.gram_mode == 0); "
                           "use kv_grad / kv_grad_generic")
    X1, Xc, unsort, _, n_c = gram_operands(x1, x2, KV_GRAM)   # (the derivative kernel centres 128-row blocks: compact AND medium rows)
    n_rows = x1.n
    wide = None
    # far-pair culling (settings.far_pair_cutoff) rides on the block-centred mode -- rows in curve order with chunk spheres; a cloud narrow enough for
    # the cloud-centred expansion has nothing far: the contracted cloud and the right vectors go in x2's Hilbert order too (the sums are order-free)
    sq = far_cull(x1, x2) if unsort is not None else None
    X2 = x2.xp
    if sq is not None:
        rt = rt if rt.dtype == torch.float32 else rt.to(torch.float32)
        rt_orig = rt
        X2, rt, sv2 = far_sorted_right(x2, rt)
    if unsort is not None:
        # block-centred expansion: the left vectors follow x1's Hilbert order (one gather of t x n floats per backward pass)
        lt = lt[:, : x1.n].index_select(1, x1.sorted_view().perm).contiguous()
        if n_c < x1.n:
            # the rows of the WIDE groups (sparse tails, SortedView: block radius beyond the 2e-5 policy of the expansion) take the
            # direct-difference row-block path, as in the forward product (kv_partials_sorted); the Gram-form kernel sees the compact rows only
            xw = PreparedPoints(X1[n_c:].contiguous(), x1.n - n_c, x1.d, x1.dp, x1.kind, x1.param)
            ltw = lt[:, n_c:].contiguous()
            if ltw.shape[1] % 4:
                ltw = torch.nn.functional.pad(ltw, (0, 4 - ltw.shape[1] % 4))
            if want_gz1 or x1.kind == "rq":
                # input gradients / the shape-parameter sum of the tail rows: dense row blocks (HIP generation + library GEMMs, float64 sums)
                wide = kv_grad_generic(xw, x2, ltw, rt if sq is None else rt_orig, want_gz1=want_gz1)
            else:
                # hyper-parameter sums only: the fused direct-difference derivative kernel (kv_grad.hpp) on the rectangular block
                sv1 = x1.sorted_view()
                far = None if sq is None else (sq, sv1.centers[n_c // 128 :], sv1.radii[n_c // 128 :], X2, rt, sv2)
                gw = kv_grad(xw, x2, ltw, rt, iso=False, far=far)
                wide = torch.cat([gw.double(), torch.zeros(1, device=dev, dtype=torch.float64)])
            lt = lt[:, :n_c].contiguous()
            n_rows = n_c
        if lt.shape[1] % 4:
            lt = torch.nn.functional.pad(lt, (0, 4 - lt.shape[1] % 4))
        nd = int(L.gpamd_kv_grad2_workspace_doubles(n_rows, x2.n, t, x1.d))
        ws = torch.empty(nd, device=dev, dtype=torch.float64)
        if want_gz1:
            nx = int(L.gpamd_kv_grad2_xworkspace_floats(n_rows, x2.n, t, x1.d))
            xws = torch.empty(nx, device=dev, dtype=torch.float32)
    # W = L^T R on the f16 matrix pipe at f32 accuracy (kv_grad2.hpp WSPLIT; the same switch as the K*V contraction) from 24 columns on: a
    # 32 x 32 tile costs 15 f16 MFMAs (480 cycles) whatever t, the fp32 form t / 2 MFMAs of 64 cycles -- cheaper below 15 columns, and exact
    # to 1e-10 of sum |W dK| where the f16 accumulation leaves a 1.5e-8 floor (measured: profiles/r03_s15_grad_split_error_floor.txt; it only
    # shows where the sum cancels by > 1e4, e.g. the gradient of a few-column solve).  The per-dimension mode keeps 40 more registers live and
    # spills from d = 8 on: those shapes stay on the fp32-MFMA contraction too
    # the per-dimension mode beyond 6 dimensions runs the split form at ONE wave per SIMD (kv_grad2.hpp g2_waves: its registers do not fit two),
    # spill-free: n = 500 000, 65 columns (profiles/r04_s20_grad_ard_highdim_one_wave.json) Matern-5/2 d = 10 483 ms (fp32 W 634, the spilling
    # two-wave build 921), RBF d = 8 338 (552), d = 16 465 (1543)
    split = _split_on() and t >= GRAD_SPLIT_MIN_COLS and (x1.d <= GRAD_SPLIT_MAX_ARD_DIM or (iso and not want_gz1))
    sws, ns = None, 0
    if split:
        ns = int(L.gpamd_kv_grad2_split_workspace_floats(n_rows, x2.n))
        sws = torch.empty(ns, device=dev, dtype=torch.float32)
        if lt.stride(0) % 4 or lt.data_ptr() % 16:
            lt = torch.nn.functional.pad(lt, (0, (-lt.shape[1]) % 4)).contiguous()
        if rt.stride(0) % 4 or rt.data_ptr() % 16:
            rt = torch.nn.functional.pad(rt, (0, (-rt.shape[1]) % 4)).contiguous()
    if sq is None:
        check(
            L.gpamd_kv_grad2_f32(
                *kind_args(x1), _ptr(X1), n_rows, _ptr(X2), x2.n, x1.d, _ptr(Xc), _ptr(lt), lt.stride(0), _ptr(rt), rt.stride(0), t,
                1 if iso else 0, _ptr(out), _ptr(gzt), ldg, _ptr(ws), nd, _ptr(xws), nx, KV_SPLIT if split else 0, _ptr(sws), ns, _stream(dev),
            ),
            "kv_grad2",
        )
    else:
        sv1 = x1.sorted_view()
        tws = _far_tile_ws(dev, int(L.gpamd_kv_grad2_far_workspace_ints(n_rows, x2.n)))
        check(
            L.gpamd_kv_grad2_far_f32(
                *kind_args(x1), _ptr(X1), n_rows, _ptr(X2), x2.n, x1.d, _ptr(Xc), _ptr(lt), lt.stride(0), _ptr(rt), rt.stride(0), t,
                1 if iso else 0, _ptr(out), _ptr(gzt), ldg, _ptr(ws), nd, _ptr(xws), nx, KV_SPLIT if split else 0, _ptr(sws), ns, _stream(dev),
                _ptr(sv1.centers), _ptr(sv1.radii), _ptr(sv2.centers), _ptr(sv2.radii), float(sq), _ptr(tws), tws.numel(),
            ),
            "kv_grad2 (far-pair culling)",
        )
    if wide is not None:
        gw, gzw = wide if want_gz1 else (wide, None)
        gw = gw.to(out.dtype)
        if iso and not want_gz1:   # MODE 0 convention: out[1] holds the single-lengthscale sum
            out[0] += gw[0]
            out[1] += gw[1 : 1 + x1.d].sum()
            out[1 + x1.dp] += gw[1 + x1.dp]
        else:
            out += gw
        if gzt is not None:
            gzt[:, n_c : x1.n] = gzw.t().to(gzt.dtype)
    if unsort is not None and gzt is not None:
        gzt = gzt.index_select(1, unsort)
    if iso and want_gz1:  # the kernel ran in per-dimension mode: fold to the single-lengthscale convention of kv_grad
        out = torch.cat([out[:1], out[1 : 1 + x1.d].sum().reshape(1), torch.zeros(x1.dp - 1, device=dev), out[1 + x1.dp :]])
    return out, (None if gzt is None else gzt[:, : x1.n].t().contiguous())


def prep_coef(kind: str) -> float:
    """z = coef * (x - shift) / lengthscale (prep_points): sqrt(log2(e)/2) for RBF, sqrt(2 nu) for Matern."""
    return {"rbf": RBF_PREP_COEF, "matern12": 1.0, "matern32": math.sqrt(3.0), "matern52": math.sqrt(5.0)}[kind]


def prep_coef_of(xp: PreparedPoints) -> float:
    """The same for a prepared cloud (RQ: 1 / sqrt(2 alpha))."""
    return 1.0 / math.sqrt(2.0 * xp.param) if xp.kind == "rq" else prep_coef(xp.kind)


def kv_grad_generic(x1: PreparedPoints, x2: PreparedPoints, lt: torch.Tensor, rt: torch.Tensor, want_gz1: bool = False):
    """Generic-path (float64, or d > 16) twin of :func:`kv_grad`; same return convention, float64 [2 + dp] (the last element is the
    shape-parameter sum  sum W dk/dp|_s  of a parametrised family -- RQ's alpha --, zero otherwise: as :func:`kv_grad2`).

    Per row block: W = left^T right (library GEMM) -> HIP ``gpamd_kernel_grad_block`` turns it into A = W * dk/ds and
    accumulates sum W*k -> the per-dimension sums  sum_ij A_ij (z_iq - z_jq)^2  expand into row sums, column sums and
    one A @ z2 product, accumulated in float64."""
    _require_gpu(lt, "left")
    assert x1.kind == x2.kind and x1.dp == x2.dp and x1.dtype == x2.dtype and lt.shape[0] == rt.shape[0]
    n, m, dp, dt = x1.n, x2.n, x1.dp, x1.dtype
    dev = lt.device
    fn = lib().gpamd_kernel_grad_block_f64 if dt == torch.float64 else lib().gpamd_kernel_grad_block_f32
    acc = torch.zeros(2, device=dev, dtype=torch.float64)
    gq = torch.zeros(dp, device=dev, dtype=torch.float64)
    cs = torch.zeros(m, device=dev, dtype=torch.float64)
    z1, z2 = x1.xp.to(torch.float64), x2.xp.to(torch.float64)
    r_all = rt[:, :m].to(dt)
    rows = int(max(1, min(n, 65535, (1 << 26) // max(m, 1))))
    st = _stream(dev)
    gz = torch.empty(n, x1.d, device=dev, dtype=torch.float64) if want_gz1 else None
    for r0 in range(0, n, rows):
        nr = min(rows, n - r0)
        w = (lt[:, r0 : r0 + nr].to(dt).t() @ r_all).contiguous()
        check(fn(*kind_args(x1), _ptr(x1.xp), r0, nr, _ptr(x2.xp), m, dp, _ptr(w), w.stride(0), _ptr(acc), st), "kernel_grad_block")
        a = w.to(torch.float64)
        zb = z1[r0 : r0 + nr]
        rs, az = a.sum(1, keepdim=True), a @ z2
        gq += (zb.pow(2) * rs).sum(0) - 2.0 * (zb * az).sum(0)
        cs += a.sum(0)
        if want_gz1:   # d/dz_i of sum_j A_ij |z_i - z_j|^2 = 2 (z_i rowsum_i - (A z2)_i): the convention of kv_grad2's Gz1
            gz[r0 : r0 + nr] = 2.0 * (zb * rs - az)[:, : x1.d]
    gq += (z2.pow(2) * cs.unsqueeze(-1)).sum(0)
    out = torch.cat([acc[:1], gq, acc[1:]])
    return (out, gz) if want_gz1 else out


# d s / d l factors: s = sum_q z_q^2-differences with z = coef * x / l  =>  ds_q/dl_q = -2 s_q / l_q
RBF_PREP_COEF = math.sqrt(0.5 * 1.4426950408889634)
