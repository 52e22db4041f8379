"""CPU emulation of the split-operand contraction (gpytorch_amd/csrc/kv_gramh.hpp, kv_vsplit.hpp): the same operand scheme --
K generated as 2^12 K and split hi = f16 toward zero, lo = f16(K - hi); every column of V scaled by a power of two so that
max |V_c| lands in [2^13, 2^14) and split with round-to-nearest; three exact f16 x f16 products accumulated in float32; the
result multiplied by 2^-12 / scale_c -- against float64.  Pins the accuracy claim of DESIGN.md 3.1b without a GPU: the split
contraction is as accurate as a plain float32 matrix product of the same operands (the 2e-5 bound of the GPU tests is dominated
by the squared distances and the float32 accumulation, which both contraction paths share), for well- and badly-scaled columns,
and documents what the 2^12 scale of K buys (it removes the one-sided truncation of small K entries at no cost)."""

import numpy as np
import pytest
import torch

KSHIFT, VEXP = 12, 14


def _rtz_f16(x: torch.Tensor) -> torch.Tensor:
    """float32 -> float16 toward zero (v_cvt_pkrtz_f16_f32) for x >= 0."""
    h = x.to(torch.float16)
    over = h.to(torch.float32) > x
    bits = h.view(torch.int16) - over.to(torch.int16)        # previous representable value
    return bits.view(torch.float16)


def _split_k(k: torch.Tensor, shift: int):
    ks = k * (2.0 ** shift)
    hi = _rtz_f16(ks)
    lo = _rtz_f16(ks - hi.to(torch.float32))
    return hi, lo


def _split_v(v: torch.Tensor):
    """v: [t, m] float32 -> (hi, lo, colmul); column = one row here (probe-major)."""
    mx = v.abs().amax(dim=1)
    scale = torch.ones_like(mx)
    nz = mx > 0
    ex = torch.frexp(mx[nz])[1]                               # mx = f 2^ex, f in [0.5, 1)
    scale[nz] = torch.ldexp(torch.ones_like(mx[nz]), (VEXP - ex).clamp(-100, 100))
    vs = v * scale[:, None]
    hi = vs.to(torch.float16)
    lo = (vs - hi.to(torch.float32)).to(torch.float16)
    return hi, lo, (2.0 ** -KSHIFT) / scale


def _split_product(k: torch.Tensor, v: torch.Tensor, shift: int = KSHIFT) -> torch.Tensor:
    kh, kl = _split_k(k, shift)
    vh, vl, colmul = _split_v(v)
    colmul = colmul * (2.0 ** (KSHIFT - shift))
    f = torch.float32
    acc = vl.to(f) @ kh.to(f) + vh.to(f) @ kl.to(f) + vh.to(f) @ kh.to(f)   # [t, n]; products exact, float32 accumulation
    return acc * colmul[:, None]


def _rbf(n, m, d, ls, seed):
    g = torch.Generator().manual_seed(seed)
    x1, x2 = torch.rand(n, d, generator=g, dtype=torch.float64), torch.rand(m, d, generator=g, dtype=torch.float64)
    sq = ((x2[:, None, :] - x1[None, :, :]) ** 2).sum(-1)
    return torch.exp(-0.5 * sq / ls ** 2)                     # K[j][i], float64


@pytest.mark.parametrize("ls", [0.1, 0.25, 1.0])
def test_split_product_matches_float64(ls):
    g = torch.Generator().manual_seed(7)
    n, m, t = 64, 3000, 9
    k64 = _rbf(n, m, 3, ls, 1)
    k = k64.float()                                           # the kernel's K is a float32 value; its own error is not under test
    v = torch.randn(t, m, generator=g)
    v[0] *= 1e-25
    v[1] *= 1e25
    v[2] = 0.0
    v[3] = 1.0 + 0.1 * torch.rand(m, generator=g)             # smooth, positive: no cancellation, a biased split would show here
    v[4] = torch.sign(v[4])
    v[5] = torch.exp(6.0 * v[5])                              # entries over ~20 orders of magnitude in one column
    got = _split_product(k, v).double()
    ref = v.double() @ k.double()
    f32 = (v @ k).double()                                    # plain float32 product of the same operands
    for c in range(t):
        scale = ref[c].abs().max().clamp_min(1e-300)
        err = float((got[c] - ref[c]).abs().max() / scale)
        err32 = float((f32[c] - ref[c]).abs().max() / scale)
        assert err < max(3.0 * err32, 1e-6), (ls, c, err, err32)
    assert float(got[2].abs().max()) == 0.0


def test_the_shift_of_k_is_what_keeps_small_entries_unbiased():
    """Short lengthscale, positive V: almost all K entries are far below 1.  Without the 2^12 scale the f16 subnormal spacing
    (6e-8) truncates the small ones toward zero -- a one-sided error (1.4e-6 of the result here); with the scale, which costs
    nothing, the product is accurate to 1e-7."""
    n, m = 16, 20000
    k = _rbf(n, m, 3, 0.02, 3).float()
    v = (1.0 + 0.1 * torch.rand(1, m, generator=torch.Generator().manual_seed(4)))
    ref = (v.double() @ k.double())[0]
    with_shift = _split_product(k, v, KSHIFT).double()[0]
    without = _split_product(k, v, 0).double()[0]
    err_with = float(((with_shift - ref).abs() / ref.abs().max()).max())
    err_without = float(((without - ref).abs() / ref.abs().max()).max())
    assert err_with < 2e-7
    assert err_without > 5 * err_with


def test_rtz_helper_is_toward_zero():
    x = torch.tensor([1.0, 1.0009765625, 1.0004, 0.3333333, 4095.9, 6.0e-5, 1e-7, 0.0])
    h = _rtz_f16(x).to(torch.float32)
    assert bool((h <= x).all())
    up = torch.nextafter(_rtz_f16(x), torch.tensor(float("inf"), dtype=torch.float16)).to(torch.float32)
    assert bool((up > x)[x > 0].all())
    assert h[0] == 1.0 and h[1] == 1.0009765625 and h[2] == 1.0


def test_operand_layout_of_the_split_kernel():
    """The index algebra of kv_gramh.hpp / kv_vsplit.hpp, emulated lane by lane: the 32x32x16 MFMA computes
    D[m][n] = sum_k A[m][k] B[k][n] with lane l = (h = l >> 5, x = l & 31) supplying A[m = x][k = 8h .. 8h+7] and
    B[k = 8h .. 8h+7][n = x], and leaves lane (h, x) with D[(r & 3) + 8 (r >> 2) + 4h][x] in register r.
      * Gram MFMA (A = x_j rows, B = x_i rows): lane (h, i) register r holds K[j(r, h)][i], j(r, h) = (r & 3) + 8 (r >> 2) + 4h;
      * contraction MFMA mf = 0, 1: B slot (h, e) := the lane's own register r = 8 mf + e;  A slot (h, e) := plane position
        16 (2 b + mf) + 8h + e of the V planes, which vsplit_kernel fills with j = 16 g + (e & 3) + 8 (e >> 2) + 4h (g = 2 b + mf).
    The two MFMAs of a 32-row block b must then add up to sum_j V[c][j] K[j][i] over the block's 32 rows."""
    rng = np.random.default_rng(0)
    nblk, C, I = 3, 32, 32
    m = 32 * nblk
    K = rng.standard_normal((m, I))            # K[j][i]
    V = rng.standard_normal((C, m))            # V[c][j]
    # vsplit_kernel: plane[c][16 g + 8 h + e] = V[c][16 g + (e & 3) + 8 (e >> 2) + 4 h]
    plane = np.zeros_like(V)
    for g in range(m // 16):
        for h in range(2):
            for e in range(8):
                plane[:, 16 * g + 8 * h + e] = V[:, 16 * g + (e & 3) + 8 * (e >> 2) + 4 * h]
    assert sorted(set((e & 3) + 8 * (e >> 2) + 4 * h for h in range(2) for e in range(8))) == list(range(16))
    out = np.zeros((C, I))
    for b in range(nblk):
        # registers of the Gram result: reg[h][i][r] = K[32 b + j(r, h)][i]
        reg = np.zeros((2, I, 16))
        for h in range(2):
            for r in range(16):
                reg[h, :, r] = K[32 * b + (r & 3) + 8 * (r >> 2) + 4 * h, :]
        for mf in range(2):
            A = np.zeros((C, 16))              # A[m = c][k]
            Bm = np.zeros((16, I))             # B[k][n = i]
            for h in range(2):
                for e in range(8):
                    A[:, 8 * h + e] = plane[:, 16 * (2 * b + mf) + 8 * h + e]
                    Bm[8 * h + e, :] = reg[h, :, 8 * mf + e]
            out += A @ Bm
    np.testing.assert_allclose(out, V @ K, rtol=0, atol=1e-12)
    # the extra (f32) column of half mf reads rows 16 mf + 4 h .. + 3 and 16 mf + 8 + 4 h .. + 3: exactly j(8 mf + 0..3, h), j(8 mf + 4..7, h)
    for mf in range(2):
        for h in range(2):
            rows = [16 * mf + 4 * h + q for q in range(4)] + [16 * mf + 8 + 4 * h + q for q in range(4)]
            assert rows == [((8 * mf + e) & 3) + 8 * ((8 * mf + e) >> 2) + 4 * h for e in range(8)]


def test_sorted_view_host_logic():
    """backend.SortedView (Hilbert order, un-sort index, chunk centres, block radius) on CPU tensors: the permutation round-trips, the
    centres are the chunk means, a clustered cloud gets compact blocks while the same points in random order do not."""
    from gpytorch_amd import backend as B

    g = torch.Generator().manual_seed(0)
    n, d = 5000, 3
    z = torch.rand(n, d, generator=g) * 40.0
    xp = torch.zeros(n, 4)
    xp[:, :d] = z
    pp = B.PreparedPoints(xp, n, d, 4, "rbf")
    sv = pp.sorted_view()
    assert torch.equal(sv.xs, xp[sv.perm])
    ld = B.round_up(n, 4)
    v = torch.randn(2, ld, generator=g)
    v_sorted = torch.zeros_like(v)
    v_sorted[:, :n] = v[:, :n][:, sv.perm]
    assert torch.equal(v_sorted.index_select(1, sv.inv_pad)[:, :n], v[:, :n])
    assert torch.allclose(sv.centers[3], sv.xs[3 * 128 : 4 * 128].mean(0))
    # every compact group respects the radius bound; wide groups (if any) sit behind the compact region
    assert sv.r2 <= B.GRAM_MAX_BLOCK_SQRADIUS and sv.n_compact % 512 == 0 or sv.n_compact == n
    dense = torch.rand(200_000, 3, generator=g) * 17.0          # U[0,1]^3 at lengthscale 0.05: every 512-point run is compact
    xpd = torch.zeros(200_000, 4)
    xpd[:, :3] = dense
    svd = B.PreparedPoints(xpd, 200_000, 3, 4, "rbf").sorted_view()
    assert svd.n_compact == 200_000 and svd.r2 < B.GRAM_MAX_BLOCK_SQRADIUS
    gauss = torch.randn(100_001, 3, generator=g) * 2.83         # N(0, 1) inputs at lengthscale 0.3: the tails are wide groups
    xpg = torch.zeros(100_001, 4)
    xpg[:, :3] = gauss
    svg = B.PreparedPoints(xpg, 100_001, 3, 4, "rbf").sorted_view()
    assert 0 < 100_001 - svg.n_compact < 0.2 * 100_001 and svg.n_compact % 512 == 0
    # (round 5) three regions: compact 512-row groups | medium 128-row chunks (within the policy against their OWN centre) | wide chunks
    assert svg.n_compact <= svg.n_block <= 100_001 and svg.n_block % 128 == 0 and 100_001 - svg.n_block < 100_001 - svg.n_compact
    assert sorted(svg.perm.tolist()) == list(range(100_001))
    # compact region: the radius bound, recomputed here for the 512-row blocks with the kernels' centre formula
    blk = svg.xs[: svg.n_compact].reshape(-1, 512, 4)
    cen = svg.centers[: svg.n_compact // 128].reshape(-1, 4, 4).mean(1, keepdim=True)
    assert float((blk - cen).pow(2).sum(-1).max()) <= B.GRAM_MAX_BLOCK_SQRADIUS * (1 + 1e-5)
    # medium region: every 128-row chunk against its own centre; and it is not empty of chunks whose 512-row run would have failed
    blk = svg.xs[svg.n_compact : svg.n_block].reshape(-1, 128, 4)
    cen = svg.centers[svg.n_compact // 128 : svg.n_block // 128].unsqueeze(1)
    assert float((blk - cen).pow(2).sum(-1).max()) <= B.GRAM_MAX_BLOCK_SQRADIUS * (1 + 1e-5)
    # wide region: every chunk there fails the 128-row test (nothing admissible was left behind), except possibly the ragged last chunk
    nw = (100_001 - svg.n_block) // 128
    blk = svg.xs[svg.n_block : svg.n_block + 128 * nw].reshape(-1, 128, 4)
    cen = svg.centers[svg.n_block // 128 : svg.n_block // 128 + nw].unsqueeze(1)
    assert float((blk - cen).pow(2).sum(-1).max(1).values.min()) > B.GRAM_MAX_BLOCK_SQRADIUS
    # points along a curve (a road network is locally one-dimensional): 512-row runs are elongated, 128-row chunks are not -> medium, not wide
    tt = torch.linspace(0, 1, 40_000, dtype=torch.float64)
    curve = torch.stack([800.0 * tt, 20.0 * torch.sin(7.0 * tt), torch.zeros_like(tt)], -1).float()
    xpc = torch.zeros(40_000, 4)
    xpc[:, :3] = curve[torch.randperm(40_000, generator=g)]
    svc = B.PreparedPoints(xpc, 40_000, 3, 4, "matern52").sorted_view()
    assert svc.n_block - svc.n_compact > 0.4 * 40_000 and svc.n_block > 0.75 * 40_000   # (the Hilbert order does not follow the curve everywhere)
    with __import__("warnings").catch_warnings():
        __import__("warnings").simplefilter("error")
        assert B.gram_mode(B.PreparedPoints(xpc, 40_000, 3, 4, "matern52"), B.PreparedPoints(xpc, 40_000, 3, 4, "matern52")) in (0, 2)
    # Hilbert order: consecutive cells of a full grid are face neighbours (no jumps, unlike the Z-order curve)
    grid = torch.stack(torch.meshgrid(*[torch.arange(8.0)] * 3, indexing="ij"), -1).reshape(-1, 3)
    pg = grid[B.hilbert_order(grid, bits=3)]
    assert float((pg[1:] - pg[:-1]).abs().sum(-1).max()) == 1.0


def test_round_down_accumulation_leaves_a_floor_that_row_signs_remove():
    """Round 4 (kv_wsplit.hpp, ws_rowsign): the split W = L^T R contraction in float64 emulation.
    * accumulation rounded to NEAREST: the error of sum_ij W_ij K_ij is ~1e-11 of sum |W_ij| K_ij -- the algorithm has no floor;
    * accumulation rounded DOWN (toward -infinity: what the f16 matrix pipe was measured to do): every W_ij is low by ~half an ulp whatever its
      sign, the error is ~4e-8 of sum |W K| however strongly the signed sum cancels -- the floor seen on the device (0.9 .. 1.5e-8);
    * the same with row i of L multiplied by a pseudo-random sign s_i before the split and the row sums multiplied by s_i afterwards: the bias
      becomes a random walk over the rows, >= 10 x smaller at n = 1500 (1 / sqrt(n) of the coherent sum)."""
    import numpy as np

    rng = np.random.default_rng(0)
    n = 1500
    x = rng.random((n, 3))
    K = np.exp(-0.5 * ((x[:, None, :] - x[None, :, :]) ** 2).sum(-1) / 0.25 ** 2)

    def split(v):
        s = 2.0 ** (12 - np.ceil(np.log2(np.abs(v).max())))
        vs = (v * s).astype(np.float32)
        hi = vs.astype(np.float16)
        lo = (vs - hi.astype(np.float32)).astype(np.float16)
        return hi.astype(np.float64), lo.astype(np.float64), s

    def rne(a):
        return a.astype(np.float32).astype(np.float64)

    def rdn(a):
        f = a.astype(np.float32)
        return np.where(f.astype(np.float64) > a, np.nextafter(f, np.float32(-np.inf)), f).astype(np.float64)

    def contract(L, R, rnd):
        Lh, Ll, sl = split(L)
        Rh, Rl, sr = split(R)
        acc = rnd(np.outer(Lh, Rl))               # the kernel's order: the two small products first, then the leading one
        acc = rnd(acc + np.outer(Ll, Rh))
        acc = rnd(acc + np.outer(Lh, Rh))
        return acc / (sl * sr)

    a = rng.standard_normal(n)
    L, R = -a, a                                   # the data-fit term of the MLL backward: the same vector on both sides
    Wt = np.outer(L.astype(np.float32).astype(np.float64), R.astype(np.float32).astype(np.float64))
    mag = (np.abs(Wt) * K).sum()
    assert mag / abs((Wt * K).sum()) > 50          # a cancelling sum
    e_rne = abs(((contract(L, R, rne) - Wt) * K).sum()) / mag
    e_rdn = abs(((contract(L, R, rdn) - Wt) * K).sum()) / mag
    sgn = np.where(rng.random(n) < 0.5, -1.0, 1.0)
    Wf = contract(L * sgn, R, rdn) * sgn[:, None]  # flipped rows in, signs taken back on the row sums
    e_flip = abs(((Wf - Wt) * K).sum()) / mag
    assert e_rne < 2e-9, e_rne
    assert 1e-8 < e_rdn < 1e-7, e_rdn
    assert e_flip < 0.1 * e_rdn, (e_flip, e_rdn)
    # ... and with independent signs on 32-row blocks of the RIGHT block as well (ws_blocksign): a random walk over n * n / 32 pairs
    sblk = np.repeat(np.where(rng.random((n + 31) // 32) < 0.5, -1.0, 1.0), 32)[:n]
    Wfb = contract(L * sgn, R * sblk, rdn) * sgn[:, None] * sblk[None, :]
    e_flip2 = abs(((Wfb - Wt) * K).sum()) / mag
    assert e_flip2 < 0.5 * max(e_flip, 2e-10) or e_flip2 < 5e-10, (e_flip2, e_flip)


def test_round4_settings_scopes_restore_their_state():
    """The three knobs round 4 added are process-global settings like the reference's: scopes nest and restore."""
    from gpytorch_amd import settings as S

    assert S.rhs_refinement.off() and S.rhs_refinement.steps == 1
    with S.rhs_refinement(True):
        assert S.rhs_refinement.on()
        with S.rhs_refinement(False):
            assert S.rhs_refinement.off()
        assert S.rhs_refinement.on()
    assert S.rhs_refinement.off()
    assert S.sharding.mll_row_group() is None
    tok = object()
    with S.sharding(probe_group=None, row_group=None, mll_row_group=tok):
        assert S.sharding.mll_row_group() is tok
    assert S.sharding.mll_row_group() is None and S.sharding.probe_group() is None
    assert S.batched_small_members.max_size == 3000 and S.batched_small_members.on()


def test_rhs_refinement_loop_squares_the_accuracy_of_a_float32_solve():
    """``bbmm.refine_with_`` (the loop behind ``settings.rhs_refinement``) on the CPU: a float32 CG solve of an ill-conditioned kernel system
    (kappa ~ 1e6) stalls at 1e-4 .. 1e-3; one float64 residual + one more float32 solve takes it below 1e-6."""
    import pytest

    try:
        from gpytorch_amd.bbmm import refine_with_
    except Exception as exc:      # the package imports its ctypes binding lazily; a missing library must not fail a CPU-only host-logic test
        pytest.skip(f"gpytorch_amd not importable here: {exc}")
    g = torch.Generator().manual_seed(0)
    n = 600
    X = torch.rand(n, 2, generator=g, dtype=torch.float64)
    A = torch.exp(-0.5 * torch.cdist(X, X).pow(2) / 0.3 ** 2) * 50.0 + 1e-4 * 50.0 * torch.eye(n, dtype=torch.float64)   # kappa ~ 1e6
    y = torch.randn(1, n, generator=g, dtype=torch.float64)
    A32 = A.float()

    def cg32(r, iters=4000):
        b = r.float().reshape(-1)
        x = torch.zeros_like(b)
        res = b.clone()
        p = res.clone()
        rs = res @ res
        for k in range(iters):
            Ap = A32 @ p
            al = rs / (p @ Ap)
            x += al * p
            res -= al * Ap
            rs2 = res @ res
            if rs2.sqrt() < 1e-6 * b.norm():
                break
            p = res + (rs2 / rs) * p
            rs = rs2
        return x.reshape(1, -1), k + 1

    sol, _ = cg32(y)
    exact = torch.linalg.solve(A, y.reshape(-1))
    e0 = float((sol.double().reshape(-1) - exact).norm() / exact.norm())
    extra = refine_with_(y, sol, lambda v: (A @ v.reshape(-1)).reshape(1, -1), cg32, steps=1)
    e1 = float((sol.double().reshape(-1) - exact).norm() / exact.norm())
    assert extra > 0
    assert e0 > 1e-5, e0                   # float32 CG alone is limited by kappa * eps
    assert e1 < 0.05 * e0, (e0, e1)        # one refinement step: more than an order of magnitude


def test_variational_quadratic_form_is_second_order_in_the_solve_error():
    """``bbmm.variational_inv_quad`` (the predictive variance of f under ``settings.rhs_refinement``, round 6): B^T A^-1 B from float32-accurate
    solves X to SECOND order in their error with ONE float64 product A X and no second solve -- X^T (2 B - A X) -- against the plain
    contraction B^T X, on an ill-conditioned kernel system (kappa ~ 1e6) whose diagonal 1 - b^T A^-1 b sits at 1e-4 .. 1e-2."""
    import pytest

    try:
        from gpytorch_amd.bbmm import variational_inv_quad
    except Exception as exc:
        pytest.skip(f"gpytorch_amd not importable here: {exc}")
    g = torch.Generator().manual_seed(3)
    n, m = 700, 37
    X = torch.rand(n, 2, generator=g, dtype=torch.float64)
    Xs = torch.rand(m, 2, generator=g, dtype=torch.float64)
    k = lambda a, c: torch.exp(-0.5 * torch.cdist(a, c).pow(2) / 0.3 ** 2)     # noqa: E731
    A = k(X, X) + 1e-4 * torch.eye(n, dtype=torch.float64)
    Bm = k(X, Xs)                                                               # [n, m] = K_X*
    exact = Bm.t() @ torch.linalg.solve(A, Bm)
    # a float32-accurate solve: the exact one with a relative perturbation of 3e-4 (what float32 mBCG attains at this conditioning), float32 storage
    Xsol = torch.linalg.solve(A, Bm)
    Xsol = (Xsol + 3e-4 * Xsol.norm(dim=0) / n ** 0.5 * torch.randn(n, m, generator=g, dtype=torch.float64)).float()
    calls = []

    def matmul64(v):
        calls.append(v.shape[1])
        assert v.dtype == torch.float64
        return A @ v

    quad = variational_inv_quad(matmul64, Bm.float(), Xsol, cols=16, rows=256)
    plain = Bm.t() @ Xsol.double()
    e_plain = float((plain - exact).diagonal().abs().max())
    e_var = float((quad - exact).diagonal().abs().max())
    assert calls == [16, 16, 5]                       # one float64 product per column group, nothing else
    assert torch.allclose(quad, quad.t())
    # (the float32 STORAGE of B itself leaves ~1e-7 relative in the form: the floor of any float32-input path)
    assert e_var < 2e-6 and e_var < 0.02 * e_plain, (e_plain, e_var)
    var_f = 1.0 - exact.diagonal()
    assert float(var_f.min()) < 1e-2 and float(((1.0 - quad.diagonal()) - var_f).abs().max()) < 2e-3 * float(var_f.min()) + 2e-6


def test_hilbert_order_cache_is_keyed_by_source_and_guarded_by_a_fingerprint():
    """``backend._ORDER_CACHE`` (round 4): one Hilbert permutation per SOURCE cloud across the evaluations of a training run (a new lengthscale
    rescales the prepared points uniformly: same order), and never across different clouds that happen to re-use an address."""
    from gpytorch_amd import backend as B

    g = torch.Generator().manual_seed(1)
    n = 4096
    xa = torch.zeros(n, 4)
    xa[:, :3] = torch.rand(n, 3, generator=g) * 40.0
    xb = torch.zeros(n, 4)
    xb[:, :3] = torch.randn(n, 3, generator=g) * 9.0
    key = ("fake-address", 0, (n, 3), "cpu", torch.float32)
    B._ORDER_CACHE.pop(key, None)
    calls = []
    orig = B.hilbert_order
    B.hilbert_order = lambda z, bits=None: (calls.append(1), orig(z, bits))[1]
    try:
        pa = B.PreparedPoints(xa, n, 3, 4, "rbf")
        pa.order_key = key
        perm_a = pa.sorted_view().perm
        assert key in B._ORDER_CACHE and len(calls) == 1
        # the same cloud at another lengthscale (uniformly rescaled, shifted): the cached permutation is re-used, no second Hilbert transform
        pa2 = B.PreparedPoints(xa * 0.37 + torch.tensor([1.0, -2.0, 0.5, 0.0]), n, 3, 4, "rbf")
        pa2.order_key = key
        # (the region split -- compact | medium | wide -- is re-evaluated on the ACTUAL coordinates, so the final order may differ; the curve order is shared)
        assert sorted(pa2.sorted_view().perm.tolist()) == list(range(n)) and len(calls) == 1
        # another cloud behind the same key (address re-use): fingerprint mismatch -> its own order, and the entry is replaced
        pb = B.PreparedPoints(xb, n, 3, 4, "rbf")
        pb.order_key = key
        svb = pb.sorted_view()
        assert len(calls) == 2
        assert sorted(svb.perm.tolist()) == list(range(n)) and not torch.equal(svb.perm, perm_a)
        assert len(B._ORDER_CACHE) <= 8
    finally:
        B.hilbert_order = orig
        B._ORDER_CACHE.pop(key, None)


def test_gram_mode_policy_round5():
    """backend.gram_mode on CPU tensors: the extent limit of the block-centred expansion (saturating split norms: 2.5e7 for the exponentially decaying
    families, the f16 range 6e4 for the heavy-tailed RQ), the wide-row share it tolerates (80 %), and the cloud-centred rule it falls under first."""
    import warnings

    from gpytorch_amd import backend as B

    g = torch.Generator().manual_seed(2)
    n = 20_000
    base = torch.rand(n, 3, generator=g)

    def pp(scale, kind):
        xp = torch.zeros(n, 4)
        xp[:, :3] = (base - base.mean(0)) * scale
        return B.PreparedPoints(xp, n, 3, 4, kind, 1.5 if kind == "rq" else None)

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        B._warned_fallback.clear()
        assert B.gram_mode(pp(5.0, "rbf"), pp(5.0, "rbf")) == 1                     # max |z|^2 = 18.75: cloud-centred
        p = pp(150.0, "matern52")                                                    # (max |z1| + max |z2|)^2 ~ 67 500: beyond the old f16-range limit
        assert 60_000 < (2 * p.zmax2 ** 0.5) ** 2 < B.GRAM_MAX_EXTENT_SQ
        sv = p.sorted_view()
        assert B.gram_mode(p, p) == (2 if n - sv.n_block <= B.GRAM_MAX_WIDE_FRACTION * n else 0)
        assert B.gram_mode(pp(25.0, "matern52"), pp(25.0, "matern52")) == 2         # dense enough at this scale: block-centred
        q = pp(150.0, "rq")
        assert B.gram_mode(q, q) == 0                                                # RQ keeps the unsaturated range
        far = pp(4000.0, "rbf")                                                      # beyond even the saturating limit
        assert (2 * far.zmax2 ** 0.5) ** 2 > B.GRAM_MAX_EXTENT_SQ and B.gram_mode(far, far) == 0


def test_direct_split_kernel_layout_emulation():
    """csrc/kv_directh.hpp on the CPU, lane by lane: the layout logic of the kernel restated with numpy index arithmetic and checked against the plain
    product.  One 32 x 32 block of pairs of one wave:
      * lane (h, i = l31) owns the 16 pairs (j(r, h), i), j(r, h) = (r & 3) + 8 (r >> 2) + 4 h; half mf = r >> 3 of them are the eight B-operand slots
        k = 8 h + e of contraction MFMA mf (a 32 x 32 x 16 f16 MFMA: D[m = c][n = i] += sum_k A[c][k] B[k][i]);
      * the x_j rows are read from the TRANSPOSED LDS image Xf[k][j] one quad at a time: rows 16 mf + 8 q + 4 h .. + 3 = elements e = 4 q .. 4 q + 3;
      * the V planes hold column c at position 16 g + 8 h + e  ->  j = 16 g + (e & 3) + 8 (e >> 2) + 4 h (kv_vsplit.hpp), so the A operand of lane (m = c, h)
        for MFMA mf is the 8 consecutive halves at 16 mf + 8 h;
      * K = hi + lo (hi: f16 toward zero of 2^12 K), V = vh + vl, product = Kh Vh + Kh Vl + Kl Vh in f32, times 2^-12 / scale_c;
    result: the 32 x 32 block of K^T-contracted columns equals V K within the split's 2^-21."""
    import numpy as np

    rng = np.random.default_rng(0)
    D = 3
    zi = rng.normal(size=(32, D)).astype(np.float32) * 0.7            # the wave's 32 output rows (one row tile)
    zj = rng.normal(size=(32, D)).astype(np.float32) * 0.7            # one 32-row j block
    V = rng.normal(size=(32, 32)).astype(np.float32)                  # [c][j] 32 columns
    # ---- pre-pass: per-column scale, planes in k-slot order
    scale = np.array([2.0 ** (14 - (np.frexp(np.abs(V[c]).max())[1])) for c in range(32)], dtype=np.float32)
    planes_h = np.zeros((32, 32), dtype=np.float16)
    planes_l = np.zeros((32, 32), dtype=np.float16)
    for c in range(32):
        for g in range(2):
            for hh in range(2):
                for e in range(8):
                    j = 16 * g + (e & 3) + 8 * (e >> 2) + 4 * hh
                    v = np.float32(V[c, j] * scale[c])
                    hi = np.float16(v)
                    planes_h[c, 16 * g + 8 * hh + e] = hi
                    planes_l[c, 16 * g + 8 * hh + e] = np.float16(v - np.float32(hi))
    Xf = zj.T.copy()                                                   # transposed LDS image [k][j]
    acc = np.zeros((32, 32), dtype=np.float64)                         # acc[c][i]: what the MFMA accumulators hold (all lanes together)
    for h in range(2):
        for i in range(32):                                            # lane (h, l31 = i)
            for mf in range(2):
                b_hi = np.zeros(8, dtype=np.float16)
                b_lo = np.zeros(8, dtype=np.float16)
                for q in range(2):
                    row0 = 16 * mf + 8 * q + 4 * h
                    quad = Xf[:, row0 : row0 + 4]                      # one ds_read_b128 per dimension
                    for el in range(4):
                        e = 4 * q + el
                        s = np.float32(((zi[i] - quad[:, el]) ** 2).sum())
                        k = np.float32(np.exp2(np.float32(12.0) - s))  # RBF, 2^KSHIFT K
                        hi = np.float32(k).view(np.uint32)             # cvt_pkrtz: toward zero
                        hv = np.float16(k)
                        if np.float32(hv) > k:                         # round-to-nearest overshot: step down one f16 ulp
                            hv = np.nextafter(hv, np.float16(0))
                        b_hi[e] = hv
                        b_lo[e] = np.float16(k - np.float32(hv))
                        # the j this slot stands for must be the j of the V plane slot the MFMA pairs it with
                        r = 8 * mf + e
                        assert (r & 3) + 8 * (r >> 2) + 4 * h == row0 + el
                # MFMA mf: D[c][i] += sum_{k = 8 h + e} A[c][k] B[k][i];  A of lane (c, h) = planes[c][16 mf + 8 h .. + 7]
                for c in range(32):
                    a_hi = planes_h[c, 16 * mf + 8 * h : 16 * mf + 8 * h + 8].astype(np.float32)
                    a_lo = planes_l[c, 16 * mf + 8 * h : 16 * mf + 8 * h + 8].astype(np.float32)
                    bh, bl = b_hi.astype(np.float32), b_lo.astype(np.float32)
                    acc[c, i] += float((a_lo * bh).sum()) + float((a_hi * bl).sum()) + float((a_hi * bh).sum())
    got = acc * (2.0 ** -12 / scale.astype(np.float64))[:, None]
    S = ((zi[None, :, :].astype(np.float64) - zj[:, None, :].astype(np.float64)) ** 2).sum(-1)    # [j][i]
    ref = V.astype(np.float64) @ np.exp2(-S)
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 2e-6, err
