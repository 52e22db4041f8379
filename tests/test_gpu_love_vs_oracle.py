"""LOVE (``fast_pred_var``) variance error against the RANK of the Lanczos decomposition: the device path and the float64 oracle on the
SAME start vector at a size the oracle can run (n = 20 000; RBF, d = 3, lengthscale 0.25, noise 0.1 -- C2's model).

Round 4 measured the reference's criterion (|variance error| < 5 % of the noise, ``test/examples/test_simple_gp_regression.py:436-442``)
missed at the reference-default rank 100 at C2 and called it "the algorithm's rank, not the kernels".  This file is the evidence: with the
same start vector the error-vs-rank curve of the HIP recurrence (float32 vectors, fused K*V, ``gpamd_lanczos_*`` kernels) coincides with
the curve of ``oracle/lanczos.py`` (float64, dense matrix) at ranks 50 / 100 / 200 / 400; and the BLOCK form that now builds the cache
(``lanczos.block_lanczos_steps``: rank = steps x 8 from eight-column products) lies on the same curve as its float64 restatement and as
the single-vector recurrence of equal rank.  Also here: the reference's multi-vector interface
(``gpytorch.root_inv_decomposition(initial_vectors [n, b], test_vectors)``, ``gpytorch/__init__.py:190-216``) through the operator on
the device against the oracle's restatement of it -- same decomposition chosen, same root.
Numbers -> gpurun_out/love_vs_oracle.json (copied into profiles/)."""
import json
import os
import time

import pytest
import torch

from tests import dense_device as DD
from tests.test_gpu_dense_at_size import synth

pytestmark = pytest.mark.gpu

N, NS, LS, THETA, S2 = 20_000, 400, 0.25, 1.0, 0.1
RANKS = (50, 100, 200, 400)


class _DenseOperator:
    """The oracle's OPERATOR: the dense float64 (or float32) matrix, held on the device and applied there with a library GEMM to the host vectors
    the oracle hands over -- the oracle's own algebra (recurrences, orthogonalisation, tridiagonal matrices) stays on the host in its own
    precision.  (On the host these products -- 400 passes over 3.2 GB -- were 70 of the module's 100 seconds.)"""

    def __init__(self, mat):
        self.mat = mat

    def __matmul__(self, v):
        return (self.mat @ v.to(self.mat.device, self.mat.dtype)).cpu()

    def float(self):
        return _DenseOperator(self.mat.float())


def _problem(dev, kind="rbf", d=3, ls=LS, n=N):
    X, y = synth(n, d)
    Xs, _ = synth(NS, d, seed=3)
    gp = DD.DenseGP(kind, X, y, ls, THETA, S2, dev)
    ks = DD.cross_rows(kind, X, Xs, ls, THETA, dev)                       # [ns, n] float64
    exact_q = (ks * gp.solve(ks.t().contiguous()).t()).sum(-1).cpu()     # k_*^T K_hat^-1 k_*
    mean_ref = (ks @ gp.alpha).squeeze(-1).cpu()                          # K_*X K_hat^-1 y (zero prior mean)
    gp.free()
    Khat = _DenseOperator(DD.dense_khat(kind, X, ls, THETA, S2, dev))    # the oracle's operator: dense float64
    return {"X": X, "y": y, "Xs": Xs, "ks": ks.cpu(), "exact_q": exact_q, "mean_ref": mean_ref, "Khat": Khat}


@pytest.fixture(scope="module")
def problem(dev):
    return _problem(dev)


def _errors(Qrows, T, ks, exact_q, ranks, block=1):
    """max |k_*^T (Q_k T_k^-1 Q_k^T) k_* - exact| / noise for the leading rank-k part of one decomposition (Lanczos is nested: the first
    k rows of Q and the leading k x k block of T ARE the k-step decomposition)."""
    out = []
    P = ks @ Qrows.t()                                                    # [ns, m]
    for k in ranks:
        k = (k // block) * block
        evals, evecs = torch.linalg.eigh(T[:k, :k])
        keep = evals > 0
        W = P[:, :k] @ (evecs[:, keep] / evals[keep].sqrt())
        out.append(float(((W**2).sum(-1) - exact_q).abs().max() / S2))
    return out


def test_love_error_vs_rank_matches_the_oracle_on_the_same_start_vector(dev, problem):
    from gpytorch_amd import backend as B
    from gpytorch_amd import lanczos as LZ
    from oracle import lanczos as OL

    X, ks, exact_q, Khat = problem["X"], problem["ks"], problem["exact_q"], problem["Khat"]
    g = torch.Generator().manual_seed(11)
    init = torch.randn(N, 8, generator=g, dtype=torch.float64)
    xp = B.prep_points("rbf", X.to(dev), torch.tensor(LS), X.mean(0).to(dev))
    sc, s2 = torch.tensor([THETA], device=dev), torch.tensor([S2], device=dev)
    log = {"n": N, "test_points": NS, "ranks": list(RANKS), "noise": S2}

    # ---- single start vector: the reference's recurrence -------------------------------------------------------------------------------
    t0 = time.perf_counter()
    Qo, To = OL.lanczos_tridiag(lambda v: Khat @ v, max(RANKS), N, init[:, :1])
    log["oracle_seconds"] = time.perf_counter() - t0
    err_o = _errors(Qo.t(), To, ks, exact_q, RANKS)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    Qt, T = LZ.lanczos_tridiag(xp, sc, s2, max(RANKS), B.to_probe_major(init[:, :1].to(dev)))
    torch.cuda.synchronize(dev)
    log["device_seconds"] = time.perf_counter() - t0
    assert Qt.shape[0] == max(RANKS)
    err_d = _errors(Qt[:, :N].double().cpu(), T.double().cpu(), ks, exact_q, RANKS)
    log["single_vector"] = {"oracle_float64": err_o, "device_float32": err_d}

    # ---- block of eight: what builds the cache at size (no counterpart in the reference) -----------------------------------------------
    def mv(q):
        return B.kv(xp, xp, q, scale=sc, dscale=s2, vd=q)

    steps = max(RANKS) // 8
    t0 = time.perf_counter()
    Qbo, Tbo = OL.block_lanczos(lambda v: Khat @ v, steps, N, init)
    log["oracle_block_seconds"] = time.perf_counter() - t0
    err_bo = _errors(Qbo.t(), Tbo, ks, exact_q, RANKS, block=8)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    Qb, Tb = LZ.block_lanczos(mv, N, dev, steps, B.to_probe_major(init.to(dev)))
    torch.cuda.synchronize(dev)
    log["device_block_seconds"] = time.perf_counter() - t0
    Qbc = Qb[:, :N].double().cpu()
    log["block_orthogonality"] = float((Qbc @ Qbc.t() - torch.eye(Qbc.shape[0], dtype=torch.float64)).abs().max())
    log["block_T_vs_QAQt"] = float((Tb.cpu() - Qbc @ (Khat @ Qbc.t())).abs().max() / Tb.abs().max())
    err_bd = _errors(Qbc, Tb.double().cpu(), ks, exact_q, RANKS, block=8)
    log["block_of_8"] = {"oracle_float64": err_bo, "device_float32": err_bd}
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/love_vs_oracle.json", "w") as f:
        json.dump(log, f, indent=1)

    floor = 2e-3      # of the noise: float32 floor of 1 - k^T K^-1 k itself (the variance of f is 1e-4 .. 2e-3 here)
    for name, dvc, orc in (("single", err_d, err_o), ("block", err_bd, err_bo)):
        for k, a, b in zip(RANKS, dvc, orc):
            # "coincide": within 25 % of each other, or both at the float32 floor
            assert abs(a - b) <= 0.25 * max(a, b) + floor, (name, k, a, b, log)
    # the error is the ALGORITHM's: above the criterion at the reference-default rank 100, far below it at rank 400 -- on either arithmetic
    assert err_o[1] > 0.05 and err_d[1] > 0.05 and err_o[3] < 0.01 and err_d[3] < 0.01, log
    # ... and follows the rank, not the way the space was generated: block of 8 against single vector at equal rank
    for k, a, b in zip(RANKS, err_bd, err_d):
        assert a <= 2.0 * b + floor, ("block vs single", k, a, b, log)
    assert log["block_orthogonality"] < 1e-4 and log["block_T_vs_QAQt"] < 1e-4, log


def test_multi_vector_interface_on_device_selects_as_the_reference_documents(dev, problem):
    """``root_inv_decomposition(initial_vectors [n, 4], test_vectors [n, 3])`` through the model-level operator: four lock-step recurrences
    (one 4-column fused product per step), the decomposition with the smallest test-vector residual returned -- against the oracle's
    restatement of ``lanczos_tridiag(init_vecs)`` + ``_postprocess_lanczos_root_inv_decomp`` on the dense float64 matrix."""
    import gpytorch_amd as gp
    from oracle import lanczos as OL
    from tests.test_gpu_model import _model

    X, Khat = problem["X"], problem["Khat"]
    y = synth(N, 3)[1]
    g = torch.Generator().manual_seed(12)
    init = torch.randn(N, 4, generator=g, dtype=torch.float64)
    init[:, 2] = problem["ks"][:50].sum(0)            # a smooth start vector among random ones: its Krylov space solves smooth test vectors best
    test = problem["ks"][100:103].t().contiguous()    # [n, 3] columns of K_X*
    rank = 40
    _, m, lik = _model("rbf", X, y, LS, THETA, S2, dev, mean=0.0)
    with torch.no_grad(), gp.settings.max_cholesky_size(0), gp.settings.max_root_decomposition_size(rank), gp.settings.tridiagonal_jitter(0.0):
        op = lik(m(X.to(dev))).lazy_covariance_matrix.evaluate_kernel()
        R = op.root_inv_decomposition(initial_vectors=init.to(dev, torch.float32), test_vectors=test.to(dev, torch.float32)).root
        with pytest.raises(ValueError, match="test_vectors"):
            op.root_inv_decomposition(initial_vectors=init.to(dev, torch.float32))
        with pytest.raises(NotImplementedError):
            op.root_inv_decomposition(method="pivoted_cholesky")
        with pytest.raises(RuntimeError, match="cannot be multiplied"):
            op.root_inv_decomposition(initial_vectors=init[:-1].to(dev, torch.float32), test_vectors=test.to(dev, torch.float32))
    Ro, idx = OL.root_inv_decomposition_multi(lambda v: Khat @ v, N, rank, init, test)
    R = R.double().cpu()
    assert R.shape == (N, rank)
    # same decomposition: the approximate solves of the test vectors agree (float32 recurrence against float64: Krylov spaces of 40 steps)
    sd, so = R @ (R.t() @ test), Ro @ (Ro.t() @ test)
    assert float((sd - so).norm() / so.norm()) < 2e-2
    # ... and it is the chosen one, not another candidate: the residual of the returned root is the oracle's minimum over the four start vectors
    res_d = float((Khat @ sd - test).norm(dim=0).sum())
    res_all = []
    for i in range(init.shape[1]):
        Ri = OL.root_inv_decomposition(lambda v: Khat @ v, N, rank, init[:, i : i + 1])
        res_all.append(float((Khat @ (Ri @ (Ri.t() @ test)) - test).norm(dim=0).sum()))
    assert idx == min(range(len(res_all)), key=res_all.__getitem__)
    assert abs(res_d - res_all[idx]) < 1e-3 * res_all[idx], (res_d, res_all)
    assert all(abs(res_d - r) > 10 * abs(res_d - res_all[idx]) + 1e-3 * r for j, r in enumerate(res_all) if j != idx), (res_d, res_all)


def test_error_at_the_reference_default_settings_is_the_stopping_rules(dev):
    """Round 5 recorded that at the REFERENCE-DEFAULT prediction settings (rank-15 pivoted-Cholesky preconditioner, ``eval_cg_tolerance`` 0.01) the
    posterior is ~1 % off in the mean and a fraction of the noise off in the exact variance, and called it "the reference's stopping rule, not
    our kernels".  The evidence, at a size the oracle can run: the SAME settings through ``oracle/linear_cg.py`` + ``oracle/pivoted_cholesky.py``
    in float64 on the dense matrix stop at the same iteration and land at the same distance from the dense-Cholesky posterior as the device
    path (float32, fused K*V) does.  Pattern: ``test/examples/test_simple_gp_regression.py:386-442`` (exact vs fast variances against a
    tolerance); here both sides against dense float64."""
    import gpytorch_amd as g
    from oracle import exact_gp as OG
    from oracle import linear_cg as OCG
    from tests.test_gpu_model import _model

    n_ = 12_000                                          # (the host's float32 dense products set the test's time: 88 + 88 iterations x 0.6 GB)
    problem = _problem(dev, n=n_)
    X, y, Xs, ks, Khat = problem["X"], problem["y"], problem["Xs"], problem["ks"], problem["Khat"]
    nv = 8                                               # test points of the exact-variance solve (8 columns through the host's dense matrix)
    mean_ref, fvar_ref = problem["mean_ref"], (THETA - problem["exact_q"])[:nv]
    S = g.settings
    _, m, lik = _model("rbf", X, y, LS, THETA, S2, dev, mean=0.0)
    m.eval(), lik.eval()
    with torch.no_grad(), S.max_cholesky_size(0), S.fast_pred_var(False):          # everything else at the defaults the reference ships
        assert S.eval_cg_tolerance.value() == 0.01 and S.max_preconditioner_size.value() == 15
        with S.skip_posterior_variances():                                         # (the mean-cache solve alone: its iteration count is compared)
            mu_d = m(Xs.to(dev)).mean.double().cpu()
        from gpytorch_amd import linear_cg as LCG

        it_d = LCG.LAST_INFO.iterations
        m.train(), m.eval()
        var_d = m(Xs[:nv].to(dev)).variance.double().cpu()
    # the oracle under the same settings AND the same arithmetic: rank-15 preconditioner (A.3 / A.4), mBCG at tolerance 0.01 (A.2) in FLOAT32 -- the
    # reference computes in the dtype of its inputs, and float32 recurrences on a kappa ~ 1e6 system need more iterations than float64 ones (88
    # against 67 here: loss of orthogonality, not a property of either implementation) -- on the dense matrix, the preconditioner applied in
    # float64 as the device path does (linear_cg.Preconditioner.apply_)
    papply64, _, _ = OG.make_preconditioner("rbf", X.double(), LS, THETA, S2, 15)
    K32 = Khat.float()

    def papply(r):
        return papply64(r.double()).float()

    def mm(V):
        return K32 @ V

    out = OCG.linear_cg(mm, y.float().unsqueeze(-1), tolerance=0.01, preconditioner=papply, return_info=True)
    sol, info = out[0].double(), out[-1]
    mu_o = (ks @ sol).squeeze(-1)
    ksv = ks[:nv].t().contiguous()
    solv = OCG.linear_cg(mm, ksv.float(), tolerance=0.01, preconditioner=papply).double()
    var_o = THETA - (ksv * solv).sum(0)
    e_mu_d = float((mu_d - mean_ref).abs().max() / mean_ref.abs().max())
    e_mu_o = float((mu_o - mean_ref).abs().max() / mean_ref.abs().max())
    e_var_d = float((var_d - fvar_ref).abs().max() / S2)
    e_var_o = float((var_o - fvar_ref).abs().max() / S2)
    log = {"n": n_, "settings": "max_preconditioner_size 15, eval_cg_tolerance 0.01 (reference defaults), exact variance on 8 test points",
           "mean_cache_cg_iterations": {"device": it_d, "oracle": info["iters"]},
           "mean_max_err_over_max_abs_mean": {"device": e_mu_d, "oracle_float32": e_mu_o},
           "variance_max_err_over_noise": {"device": e_var_d, "oracle_float32": e_var_o}}
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/default_settings_vs_oracle.json", "w") as f:
        json.dump(log, f, indent=1)
    assert abs(it_d - info["iters"]) <= max(2, 0.1 * info["iters"]), log      # (float32 rounding moves the crossing of the tolerance by a few iterations)
    # the same distance from the truth on either implementation (within 25 %, or both under a floor of 1e-4): the error is the stopping rule's
    # (the MAXIMUM over 400 test points of the error of two different 88-step float32 iterates: the same size, not the same number -- measured 0.95 %
    # against 1.5 %; the variances below, sums over all n entries of the solves, agree to 0.4 %)
    assert 0.5 * e_mu_o - 1e-4 <= e_mu_d <= 2.0 * e_mu_o + 1e-4, log
    assert abs(e_var_d - e_var_o) <= 0.25 * max(e_var_d, e_var_o) + 2e-3, log
    assert e_mu_o > 1e-3, log      # ... and it IS an error: the reference-default tolerance is not a 1e-3 solve


def test_love_curve_on_the_c3_model_is_the_algorithms_too(dev):
    """Round 5 recorded that LOVE misses the reference's 5 %-of-noise criterion at EVERY rank on C3's model (Matern-5/2, d = 10: 4.4 / 2.2 / 1.03 x
    noise at rank 100 / 400 / 1600, n = 60 000) and had the oracle-coincidence evidence for the RBF d = 3 model only.  Same evidence for that
    model, at a size the host runs in seconds (n = 12 000): the device recurrence and ``oracle/lanczos.py`` (float64, dense matrix) on the same
    start vector give the same error at ranks 100 and 400 -- both far above the criterion: the slowly decaying Matern spectrum needs a rank LOVE
    does not reach, on any arithmetic."""
    from gpytorch_amd import backend as B
    from gpytorch_amd import lanczos as LZ
    from oracle import lanczos as OL

    kind, d, ls, n, ranks = "matern52", 10, 0.8, 12_000, (100, 400)
    pr = _problem(dev, kind, d, ls, n)
    X, ks, exact_q, Khat = pr["X"], pr["ks"], pr["exact_q"], pr["Khat"]
    init = torch.randn(n, 1, generator=torch.Generator().manual_seed(13), dtype=torch.float64)
    Qo, To = OL.lanczos_tridiag(lambda v: Khat @ v, max(ranks), n, init)
    err_o = _errors(Qo.t(), To, ks, exact_q, ranks)
    xp = B.prep_points(kind, X.to(dev), torch.tensor(ls), X.mean(0).to(dev))
    sc, s2 = torch.tensor([THETA], device=dev), torch.tensor([S2], device=dev)
    Qt, T = LZ.lanczos_tridiag(xp, sc, s2, max(ranks), B.to_probe_major(init.to(dev)))
    err_d = _errors(Qt[:, :n].double().cpu(), T.double().cpu(), ks, exact_q, ranks)
    log = {"kind": kind, "d": d, "n": n, "lengthscale": ls, "ranks": list(ranks), "variance_max_err_over_noise": {"oracle_float64": err_o, "device_float32": err_d}}
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/love_vs_oracle_c3_model.json", "w") as f:
        json.dump(log, f, indent=1)
    for k, a, b in zip(ranks, err_d, err_o):
        assert abs(a - b) <= 0.25 * max(a, b) + 2e-3, (k, a, b, log)
    assert min(err_o) > 0.05 and min(err_d) > 0.05, log      # the criterion is missed by the ALGORITHM at these ranks
