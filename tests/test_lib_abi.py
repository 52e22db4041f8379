"""CPU: libgpamd.so loads and exports every symbol declared in include/gpamd.h; host-side logic
(launch planning, tridiagonal assembly, settings) behaves -- no compute calls (no GPU here)."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "gpamd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gpamd_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    from gpytorch_amd._lib import LIB_PATH, SIGNATURES, lib

    assert os.path.exists(LIB_PATH), "run __graft_entry__.build() first"
    h = lib()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(h, name), f"{name} declared in gpamd.h but not exported"
    assert sorted(SIGNATURES) == declared, "ctypes table and gpamd.h disagree"
    assert h.gpamd_abi_version() == 5


def test_library_exports_nothing_beyond_the_header():
    """The converse: every ``gpamd_*`` symbol the product library exports is declared in include/gpamd.h (tuning /
    ablation entry points live in libgpamd_tune.so, not here)."""
    import subprocess

    from gpytorch_amd._lib import LIB_PATH

    nm = "/opt/rocm/lib/llvm/bin/llvm-nm" if os.path.exists("/opt/rocm/lib/llvm/bin/llvm-nm") else "nm"
    out = subprocess.run([nm, "-D", "--defined-only", LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("gpamd_")})
    declared = _declared_symbols()
    extra = [s_ for s_ in exported if s_ not in declared]
    assert not extra, f"exported but not declared in include/gpamd.h: {extra}"


def test_kv_plan_covers_and_fills():
    from gpytorch_amd import backend as B

    for n, m, t in [(2000, 2000, 11), (100_000, 100_000, 65), (500_000, 500_000, 65), (1_000_000, 1_000_000, 33),
                    (10_000, 100_000, 1), (257, 300, 140)]:
        for kind, flags in (("rbf", 0), ("rbf", B.KV_GRAM), ("matern52", 0)):
            S, jc, ws = B.kv_plan(kind, n, m, 3, t, flags, B.round_up(n, 4))
            assert S >= 1 and jc % 128 == 0 and S * jc >= m and (S - 1) * jc < m
            assert ws == S * t * B.round_up(n, 4)
        # split-operand contraction: the two f16 planes of V (one row per 32-column tile row, round_up(m, 128) positions) and the
        # per-column scales live behind the slabs; nothing extra below five columns (the VALU-contraction kernel runs there)
        S, jc, ws = B.kv_plan("rbf", n, m, 3, t, B.KV_GRAM | B.KV_SPLIT, B.round_up(n, 4))
        slabs = S * t * B.round_up(n, 4)
        assert S >= 1 and jc % 128 == 0 and S * jc >= m
        if t < 5:
            assert ws == slabs
        else:
            ldh = (m + 127) // 128 * 128
            full, rem = divmod(t, 64)
            rows = 64 * full + (0 if rem < 5 else 32 * ((rem - (1 if rem % 32 == 1 else 0) + 31) // 32))
            if rem == 1 and full:          # 64 k + 1 columns: the last group carries the extra column, no further plane rows
                rows = 64 * full
            assert ws >= slabs + rows * ldh and ws <= slabs + rows * ldh + 2 * (2 * t + 64) + 8 and ws % 4 == 0, (n, m, t, ws, slabs, rows)


def test_argument_validation_without_gpu():
    from gpytorch_amd._lib import lib

    h = lib()
    assert h.gpamd_kv_plan(0, 0, 10, 3, 1, 0, 12, None, None, None) == -1
    assert b"bad shape" in h.gpamd_last_error()
    # input dimensions beyond 32 are refused before any launch (ABI version 4; 16 before)
    rc = h.gpamd_kv_partials_f32(0, 0.0, None, 10, None, 10, 40, None, None, 12, 1, None, 12, 1, 128, 0, None, None)
    assert rc == -2
    # split-operand contraction: the j chunk must be the plan's (a multiple of the 128-row LDS tile) -- refused before any launch
    rc = h.gpamd_kv_partials_f32(0, 0.0, None, 1000, None, 1000, 3, None, None, 1000, 11, None, 1000, 1, 1004, 1 | 8, None, None)
    assert rc == -1 and b"jchunk % 128" in h.gpamd_last_error()
    # far-pair culling (ABI version 5): a cutoff without the spheres / the tile-list workspace is refused before any launch; the workspace bound
    rc = h.gpamd_kv_partials_far_f32(0, 0.0, None, 1000, None, 1000, 3, None, None, 1000, 11, None, 1000, 1, 1024, 1 | 8, None, None, None, None, None, None, 23.0, None, 0)
    assert rc == -1 and b"bounding-sphere" in h.gpamd_last_error()
    assert h.gpamd_kv_far_workspace_ints(1000, 3, 1024) == 8 * 3 * 9 and h.gpamd_kv_far_workspace_ints(0, 1, 128) == 0
    # batched small-member entry points: shapes and the RQ shape-parameter array are checked before any launch
    assert h.gpamd_kernel_dense_batched_f32(0, None, None, 0, None, 5, 4, 2, None, None, None, 8, None) == -1
    assert h.gpamd_kernel_dense_batched_f32(4, None, None, 5, None, 5, 4, 2, None, None, None, 8, None) == -1 and b"kparam" in h.gpamd_last_error()
    assert h.gpamd_kernel_grad_batched_f32(0, None, None, 5, None, 5, 20, 2, None, 8, None, None) == -1 and b"dp <= 16" in h.gpamd_last_error()
    # float64 / generic entry points take the shape parameter too (ABI version 3): RQ needs alpha > 0
    assert h.gpamd_prep_points_f64(4, 0.0, None, 5, 2, 2, None, 1, None, None, 4, None) == -1 and b"alpha must be positive" in h.gpamd_last_error()


def test_build_tridiag_matches_oracle():
    from gpytorch_amd.linear_cg import build_tridiag
    from oracle import exact_gp as OG
    from oracle import linear_cg as OCG
    from tests.util import make_data

    n = 200
    X, y = make_data(n, 3)
    mm = OG.make_matmul("rbf", X, 0.25, 1.0, 0.1)
    rhs = torch.randn(n, 6, dtype=torch.float64)
    for tol, max_iter in [(1.0, 1000), (1e-9, 15), (1e-3, 1000)]:
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            _, T, info = OCG.linear_cg(mm, rhs, n_tridiag=4, tolerance=tol, max_iter=max_iter, return_info=True)
        a = torch.stack(info["alpha"])[:, :4]
        b = torch.stack(info["beta"])[:, :4]
        T2 = build_tridiag(a[:20], b[:20], info["iters"], info["tolerance_reached"], 20)
        assert T2.shape == T.shape
        assert torch.allclose(T2, T, rtol=1e-12, atol=1e-12)


def test_settings_contexts():
    from gpytorch_amd import settings

    assert settings.cg_tolerance.value() == 1.0 and settings.eval_cg_tolerance.value() == 0.01
    with settings.cg_tolerance(1e-4), settings.max_cholesky_size(0), settings.fast_pred_var():
        assert settings.cg_tolerance.value() == 1e-4
        assert settings.max_cholesky_size.value() == 0
        assert settings.fast_pred_var.on()
    assert settings.cg_tolerance.value() == 1.0 and settings.max_cholesky_size.value() == 800
    assert settings.fast_pred_var.off()
    with settings.fast_computations(log_prob=False):
        assert settings.fast_computations.log_prob.off() and settings.fast_computations.solves.on()
    assert settings.fast_computations.log_prob.on()
    assert settings.min_variance.value(torch.float32) == 1e-6


def test_far_pair_cutoff_setting_and_distance():
    """settings.far_pair_cutoff: off by default (the reference's arithmetic), validated; backend.far_sq_cutoff inverts each family's profile in the
    PREPARED coordinates of csrc/common.hpp (RBF k = 2^-s, Matern k = poly(r) e^-r with r = sqrt(s), RQ k = (1 + s)^-alpha)."""
    import math

    import pytest

    from gpytorch_amd import backend as B
    from gpytorch_amd import settings

    assert settings.far_pair_cutoff.value() is None
    with settings.far_pair_cutoff(1e-7):
        assert settings.far_pair_cutoff.value() == 1e-7
    assert settings.far_pair_cutoff.value() is None
    for bad in (0.0, 1.0, -1e-3, 2.0):
        with pytest.raises(ValueError):
            settings.far_pair_cutoff(bad)
    for eps in (1e-2, 1e-5, 1e-7):
        assert abs(2.0 ** -B.far_sq_cutoff("rbf", eps) - eps) < 1e-12 * eps + 1e-18
        assert abs((1.0 + B.far_sq_cutoff("rq", eps, 1.3)) ** -1.3 - eps) < 1e-9 * eps
        for kind, poly in (("matern12", lambda r: 1.0), ("matern32", lambda r: 1.0 + r), ("matern52", lambda r: 1.0 + r + r * r / 3.0)):
            r = math.sqrt(B.far_sq_cutoff(kind, eps))
            assert abs(poly(r) * math.exp(-r) - eps) < 1e-6 * eps and poly(r * 1.001) * math.exp(-r * 1.001) < eps
    # longer tails need larger cutoffs
    assert B.far_sq_cutoff("matern12", 1e-7) < B.far_sq_cutoff("matern32", 1e-7) < B.far_sq_cutoff("matern52", 1e-7)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "gpytorch_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
                # ... nor the test-only linear_operator shim / CPU double of tests/shim (tests/test_reference_layers_cpu.py)
                assert not re.search(r"^\s*(from|import)\s+tests\b", src, flags=re.M) and "cpu_backend" not in src, f"{f} imports test infrastructure"


def test_integration_stub_matches_the_abi():
    """The binding stub shown to gpytorch maintainers in INTEGRATION.md declares the same argument lists as the
    library's own ctypes table (which test_header_symbols_exported_and_bound ties to include/gpamd.h)."""
    import ctypes as C
    import re

    from gpytorch_amd._lib import SIGNATURES

    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    names = {"_i": C.c_int, "_p": C.c_void_p, "_l": C.c_int64, "_f": C.c_float, "C.POINTER(_i)": C.POINTER(C.c_int), "C.POINTER(_l)": C.POINTER(C.c_int64)}
    found = re.findall(r"_lib\.(gpamd_\w+)\.argtypes\s*=\s*\[([^\]]*)\]", text)
    assert len(found) >= 3
    for fn, args in found:
        got = [names[a.strip()] for a in re.findall(r"C\.POINTER\(_\w\)|_\w", args)]
        want = list(SIGNATURES[fn][1])
        assert len(got) == len(want), fn
        for a, b in zip(got, want):
            # the product table writes output pointers as plain c_void_p or typed pointers: both are pointer-sized
            assert a is b or (C.sizeof(a) == C.sizeof(b) == C.sizeof(C.c_void_p)), (fn, a, b)


def test_slq_logdet_host_logic_matches_oracle():
    """bbmm.slq_logdet / lanczos.tridiag_to_diag (host side, k <= 20 eigh) against the restated StochasticLQ: random SPD
    tridiagonals, a matrix with a negative eigenvalue (masked: eigenvalue 1, eigenvector 0) and NaN propagation."""
    import math

    from gpytorch_amd.bbmm import slq_logdet
    from oracle import slq as OS

    g = torch.Generator().manual_seed(3)
    t, k, n = 5, 12, 1000
    a = 2.0 + torch.rand(t, k, generator=g, dtype=torch.float64)
    b = 0.3 * torch.rand(t, k - 1, generator=g, dtype=torch.float64)
    T = torch.diag_embed(a) + torch.diag_embed(b, 1) + torch.diag_embed(b, -1)
    assert abs(float(slq_logdet(T, n)) - float(OS.slq_logdet(T, n))) < 1e-10 * abs(float(OS.slq_logdet(T, n)))
    # exactness on a diagonal T: (n / t) * sum_j log T_j[0, 0]
    Td = torch.diag_embed(a)
    assert abs(float(slq_logdet(Td, n)) - n / t * float(a[:, 0].log().sum())) < 1e-9
    Tn = T.clone()
    Tn[0, 0, 0] = -5.0  # indefinite: the negative eigenvalue is masked, the result stays finite and equals the oracle's
    assert math.isfinite(float(slq_logdet(Tn, n)))
    assert abs(float(slq_logdet(Tn, n)) - float(OS.slq_logdet(Tn, n))) < 1e-9 * max(1.0, abs(float(OS.slq_logdet(Tn, n))))
    Tn[1, 2, 2] = float("nan")
    assert math.isnan(float(slq_logdet(Tn, n)))
