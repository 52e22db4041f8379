"""A test that could actually fail: the two kernels that carry > 95 % of all GPU time read the VGPR results of their Gram MFMAs on the VALU at a
distance the static audit can only count in wait states (``kv_gram_kernel``: 20 since round 5; ``kv_gramh_kernel``: the toolchain's 12, one
contraction MFMA among them).  Here the SAME templates are instantiated twice in ``libgpamd_tune.so`` (``csrc/tune/tune_hazard.hip``, namespace
``gpamd_hz``): as the product compiles them, and with the full data-dependent fence (32 wait states, ``common.hpp``) behind every Gram MFMA.  Both run
on one full chip -- n = 500 000 rows for the D = 1 instantiations the round-4 audit named (one Gram MFMA per block: the shortest distance), every
SIMD holding its full complement of waves -- many times over, and the partial slabs are compared BITWISE: the round-3 hazard returned
stale values in some lanes, different on every run, only on a full chip.  A single differing bit fails the test."""
import ctypes as C
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TUNE = os.path.join(ROOT, "gpytorch_amd", "csrc", "libgpamd_tune.so")

# (which, variant, kind, d, t, n, repetitions): which 0 = kv_gram_kernel, 1 = kv_gramh_kernel; variants as listed in tune_hazard.hip
CASES = [
    (0, 0, "rbf", 1, 32, 500_000, 12), (0, 1, "matern32", 1, 32, 500_000, 12), (0, 2, "rbf", 3, 65, 200_000, 10), (0, 3, "matern52", 10, 65, 120_000, 10),
    (1, 0, "matern32", 1, 64, 500_000, 25), (1, 1, "rbf", 1, 32, 500_000, 25), (1, 2, "rbf", 3, 65, 300_000, 20), (1, 3, "matern52", 10, 65, 200_000, 12),
]


@pytest.mark.skipif(not os.path.exists(TUNE), reason="libgpamd_tune.so not built (make -C gpytorch_amd/csrc tune)")
@pytest.mark.parametrize("which,variant,kind,d,t,n,reps", CASES, ids=[f"{'gramh' if c[0] else 'gram'}-{c[2]}-d{c[3]}-t{c[4]}" for c in CASES])
def test_product_code_path_is_bitwise_the_fully_fenced_build_on_a_full_chip(which, variant, kind, d, t, n, reps, dev):
    from gpytorch_amd import backend as B

    h = C.CDLL(TUNE)
    f = h.gpamd_tune_hazard_launch
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                  C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
    g = torch.Generator().manual_seed(100 * which + variant)
    X = torch.rand(n, d, generator=g).to(dev)
    ls = 0.25 if d <= 3 else 0.8
    xp = B.prep_points(kind, X, torch.tensor(ls), X.mean(0))
    assert xp.zmax2 <= B.GRAM_MAX_SQNORM
    ld = B.round_up(n, 4)
    ldh = (n + 127) // 128 * 128
    V = torch.randn(t, ld, generator=g).to(dev)
    tc = 32 * ((t - (1 if t % 32 == 1 else 0) + 31) // 32)
    Vh = (4096.0 * torch.randn(tc, ldh, generator=g)).to(dev).half()
    Vl = torch.randn(tc, ldh, generator=g).to(dev).half()
    colmul = torch.ones(tc + 1, device=dev)
    S, jc, _ = B.kv_plan(kind, n, n, d, t, B.KV_GRAM | (B.KV_SPLIT if which else 0), ld)
    P = [torch.empty(S * t * ld, device=dev), torch.empty(S * t * ld, device=dev)]
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def launch(safe, out):
        rc = f(which, variant, safe, xp.xp.data_ptr(), n, xp.xp.data_ptr(), n, V.data_ptr(), ld, t, Vh.data_ptr(), Vl.data_ptr(), ldh, colmul.data_ptr(),
               out.data_ptr(), ld, S, jc, st)
        assert rc == 0, rc

    launch(1, P[1])
    torch.cuda.synchronize(dev)
    ref = P[1].clone()
    assert bool(torch.isfinite(ref).all()) and float(ref.abs().max()) > 0
    # the fenced build reproduces itself ...
    launch(1, P[1])
    assert torch.equal(P[1], ref)
    # ... and the product's code path reproduces IT, every time
    for r in range(reps):
        P[0].zero_()
        launch(0, P[0])
        same = torch.equal(P[0], ref)
        if not same:
            bad = (P[0] != ref).nonzero().reshape(-1)
            raise AssertionError(f"repetition {r}: {bad.numel()} of {ref.numel()} slab entries differ from the fenced build (first at {int(bad[0])})")
