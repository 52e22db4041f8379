"""CPU, world_size = 2, gloo: the probe-column sharding design is exact -- two ranks that
each own half of the probes (the y column rides on rank 0 only and its solve is broadcast at the end), exchanging only
the 2-float residual statistics per iteration and the scalar SLQ sums, reproduce the single-process solve bit for bit:
same iteration count, same solutions, same log-det.  Exercises gpytorch_amd.distributed (host logic)
with the oracle standing in for the device kernels (no GPU here)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.util import free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from gpytorch_amd import distributed as D
    from oracle import exact_gp as OG
    from oracle import linear_cg as OCG
    from oracle import slq as OS
    from tests.util import make_data

    group = D.init_from_env("gloo")
    # the constructor-compatible MultiDeviceKernel: in a process group of more than one rank it installs WORLD as the probe / row group
    import gpytorch_amd as g
    from gpytorch_amd import settings as S0

    assert S0.sharding.probe_group() is None and S0.sharding.row_group() is None
    import warnings as _w

    with _w.catch_warnings(record=True) as caught:
        _w.simplefilter("always")
        mdk = g.kernels.MultiDeviceKernel(g.kernels.RBFKernel(), device_ids=[torch.device("cpu")] * world)
    assert any("installed the WORLD process group" in str(c.message) for c in caught)     # process-global state is announced ...
    # the automatic layout policy: posterior solves row-sharded over WORLD, MLL grid chosen per evaluation (2 ranks, 300 points, 6 probes: 2 x 1)
    assert S0.sharding.is_auto() and S0.sharding.row_group() is dist.group.WORLD and mdk.base_kernel.has_lengthscale
    assert S0.sharding.mll_groups(300, 6) == (dist.group.WORLD, None)
    mdk.release()                                                                          # ... and can be taken back
    assert not S0.sharding.is_auto() and S0.sharding.probe_group() is None and S0.sharding.row_group() is None
    n, t_total = 300, 6
    X, y = make_data(n, 3)
    Z = torch.randn(n, t_total, generator=torch.Generator().manual_seed(1234), dtype=torch.float64)
    Z = Z / Z.norm(dim=-2, keepdim=True)
    mm = OG.make_matmul("rbf", X, 0.25, 1.0, 0.1)
    a, b = D.probe_shard(t_total, world, rank)
    t = b - a
    rhs = torch.cat([Z[:, a:b], y.unsqueeze(-1)], dim=-1) if rank == 0 else Z[:, a:b]   # the rhs owner is rank 0

    def mean_fn(rnorm):
        return D.allreduce_residual_stats(rnorm.sum(), torch.tensor(float(rnorm.numel())), group)

    sol, T, info = OCG.linear_cg(mm, rhs, n_tridiag=t, tolerance=0.5, return_info=True, mean_residual_fn=mean_fn)
    ld = OS.slq_logdet(T, n) * (t / t_total)
    D.allreduce_sum_(ld, group)
    ysol = sol[:, -1].clone() if rank == 0 else torch.zeros(n, dtype=sol.dtype)
    D.broadcast_(ysol, 0, group)
    sol = torch.cat([sol[:, :t], ysol.unsqueeze(-1)], dim=-1)
    # the per-rank probe generator of settings.sharding: created once per (group, device), ADVANCES between evaluations (fresh probes
    # every MLL call, as on one GPU), and differs between ranks
    from gpytorch_amd import settings as S

    g1 = S.sharding.rank_generator(group, torch.device("cpu"))
    d1 = torch.randn(4, generator=g1)
    g2 = S.sharding.rank_generator(group, torch.device("cpu"))
    d2 = torch.randn(4, generator=g2)
    assert g1 is g2 and not torch.equal(d1, d2)
    q.put((rank, info["iters"], sol.numpy(), float(ld), (a, b), d1.tolist()))  # plain data: a tensor would travel as a shm handle that dies with the worker
    dist.barrier()
    dist.destroy_process_group()


def test_probe_sharding_matches_single_process():
    sys.path.insert(0, ROOT)
    from oracle import exact_gp as OG
    from oracle import linear_cg as OCG
    from oracle import slq as OS
    from tests.util import make_data

    world, port = 2, free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=180) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    n, t_total = 300, 6
    X, y = make_data(n, 3)
    Z = torch.randn(n, t_total, generator=torch.Generator().manual_seed(1234), dtype=torch.float64)
    Z = Z / Z.norm(dim=-2, keepdim=True)
    mm = OG.make_matmul("rbf", X, 0.25, 1.0, 0.1)
    # single process: all probes + y -- the sharded run averages the residual norms over exactly these columns
    rhs = torch.cat([Z, y.unsqueeze(-1)], dim=-1)
    sol, T, info = OCG.linear_cg(mm, rhs, n_tridiag=t_total, tolerance=0.5, return_info=True)
    ld = OS.slq_logdet(T, n)
    assert results[0][5] != results[1][5]                      # ranks draw different probes
    for rank, iters, s, ldr, (a, b), _ in results:
        s = torch.from_numpy(s)
        assert iters == info["iters"]
        assert torch.allclose(s[:, : b - a], sol[:, a:b], rtol=0, atol=1e-12)
        assert torch.allclose(s[:, -1], sol[:, t_total], rtol=0, atol=1e-12)
        assert abs(ldr - float(ld)) < 1e-9 * abs(float(ld))


def test_probe_shard_partition():
    sys.path.insert(0, ROOT)
    from gpytorch_amd.distributed import probe_shard

    for t_total, world in [(64, 8), (65, 8), (7, 3), (2, 4), (256, 8)]:
        spans = [probe_shard(t_total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == t_total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        probe_shard(4, 2, 2)


def _row_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from gpytorch_amd import backend as B
    from gpytorch_amd import distributed as D

    group = D.init_from_env("gloo")
    n, dp, t = 1001, 4, 3
    g = torch.Generator().manual_seed(5)
    xp = B.PreparedPoints(torch.rand(n, dp, generator=g), n, 3, dp, "rbf")
    rs = D.RowShard(xp, group)
    full = torch.zeros(t, B.round_up(n, 4))
    full[:, :n] = torch.randn(t, n, generator=g)
    loc = rs.local(full)
    back = rs.gather(loc)
    ok_gather = bool(torch.equal(back[:, :n], full[:, :n])) and bool((back[:, n:] == 0).all())
    ok_pad = bool(torch.equal(rs.x_all.xp[:n], xp.xp)) and rs.x_all.n == world * rs.n_pad and rs.x_loc.n == rs.r1 - rs.r0
    # solver partials: [t][stride] with 5 used entries per column -> entry 0 = global column sum, rest cleared
    stride = 16
    fs = torch.zeros(7 + t * stride)
    v = fs[7:].view(t, stride)
    v[:, :5] = torch.arange(t * 5, dtype=torch.float32).view(t, 5) + rank
    expect = sum((torch.arange(t * 5, dtype=torch.float32).view(t, 5) + r).sum(1) for r in range(world))
    rs.allreduce_partials(fs, 7, t, stride)
    ok_part = bool(torch.equal(v[:, 0], expect)) and bool((v[:, 1:] == 0).all())
    bc = torch.full((4,), float(rank))
    rs.broadcast(bc)
    q.put((rank, rs.r0, rs.r1, ok_gather, ok_pad, ok_part, bool((bc == 0).all())))
    dist.barrier()
    dist.destroy_process_group()


def test_row_shard_host_logic():
    """RowShard on CPU tensors over gloo, 3 ranks: the row partition covers [0, n) without overlap,
    gather(local(v)) reproduces v (zero padded tail), the replicated cloud is padded consistently, the solver's partial
    arrays are summed over ranks into entry 0, broadcast takes rank 0's value."""
    world, port = 3, free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_row_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=180) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0][1] == 0 and results[-1][2] == 1001
    for a, b in zip(results[:-1], results[1:]):
        assert a[2] == b[1]
    for r in results:
        assert all(r[3:]), r


def _row_cg_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from gpytorch_amd import backend as B
    from gpytorch_amd import distributed as D
    from oracle import kernels as OK
    from oracle import linear_cg as OCG
    from tests.util import make_data

    group = D.init_from_env("gloo")
    n, t = 403, 3
    X, y = make_data(n, 3)
    rhs = torch.cat([y.unsqueeze(-1), torch.randn(n, t - 1, generator=torch.Generator().manual_seed(8), dtype=torch.float64)], -1)
    # RowShard only needs the shapes of the prepared cloud here; the oracle's dense K stands in for the device kernels
    rs = D.RowShard(B.PreparedPoints(X.clone(), n, 3, 3, "rbf"), group)
    K_rows = OK.kernel_matrix("rbf", X[rs.r0 : rs.r1], X, 0.25, 1.0, x1_eq_x2=False)  # this rank's rows of K

    def mm_local(D_loc):  # [n_loc, t] -> this rank's rows of K_hat @ D, D gathered over ranks first
        full = rs.gather(D_loc.t().contiguous())[:, :n].t()
        return K_rows @ full + 0.1 * D_loc

    def rowsum(v):
        return D.allreduce_sum_(v.clone(), group)

    sol_loc, info = OCG.linear_cg(mm_local, rhs[rs.r0 : rs.r1], tolerance=1e-6, max_iter=300, return_info=True, rowsum_fn=rowsum)
    q.put((rank, rs.r0, rs.r1, info["iters"], sol_loc.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_row_sharded_cg_matches_single_process():
    """SURVEY.md 8e.2 on CPU, 3 ranks over gloo, float64: every rank owns a block of ROWS of K_hat, the search
    directions are all-gathered per product (RowShard.gather) and every inner product / norm of the restated
    ``linear_cg`` is all-reduced.  Same iteration count and the same solution (1e-6 of its scale) as the single-process solve."""
    sys.path.insert(0, ROOT)
    from oracle import exact_gp as OG
    from oracle import linear_cg as OCG
    from tests.util import make_data

    world, port = 3, free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_row_cg_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=180) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n, t = 403, 3
    X, y = make_data(n, 3)
    rhs = torch.cat([y.unsqueeze(-1), torch.randn(n, t - 1, generator=torch.Generator().manual_seed(8), dtype=torch.float64)], -1)
    sol, info = OCG.linear_cg(OG.make_matmul("rbf", X, 0.25, 1.0, 0.1), rhs, tolerance=1e-6, max_iter=300, return_info=True)
    for rank, r0, r1, iters, sol_loc in results:
        sol_loc = torch.from_numpy(sol_loc)
        assert iters == info["iters"]
        # (unlike probe sharding, the reduction ORDER changes: rounding differences grow along the CG recurrence)
        assert torch.allclose(sol_loc, sol[r0:r1], rtol=0, atol=1e-6 * float(sol.abs().max()))


def _kron_problem():
    from oracle import kernels as OK
    from oracle import multitask as OM

    g = torch.Generator().manual_seed(11)
    n, T = 70, 3
    X = torch.rand(n, 2, generator=g, dtype=torch.float64)
    Bf = 0.6 * torch.randn(T, 1, generator=g, dtype=torch.float64)
    v = 0.2 + 0.3 * torch.rand(T, generator=g, dtype=torch.float64)
    task_noise = 0.05 + 0.1 * torch.rand(T, generator=g, dtype=torch.float64)
    K = torch.kron(OK.kernel_matrix("rbf", X, X, 0.3, 1.2, x1_eq_x2=True), OM.task_covar(Bf, v))   # noise-free part, interleaved layout
    dvec = task_noise.repeat(n)
    y = torch.randn(n * T, generator=g, dtype=torch.float64)
    return K, dvec, y


def _kron_solve(K, dvec, y, probes, t_total, owns_rhs, mean_fn=None):
    """The structured-operator MLL ingredients with the product's host pieces (row-callback pivoted Cholesky, vector-diagonal
    preconditioner) around the oracle's CG: what KroneckerInvQuadLogdetFn runs per rank, the device product replaced by a dense one."""
    from gpytorch_amd.bbmm import build_preconditioner_rows
    from oracle import linear_cg as OCG
    from oracle import slq as OS

    N = K.shape[0]
    pre = build_preconditioner_rows(lambda p: K[p], K.diagonal().clone(), dvec, True, rank=12, tol=1e-8, min_size=0)
    ld = pre.q1t.shape[1]

    def papply(R):                                       # oracle layout (n, c) <-> the product's probe-major [c, ld]
        Rt = torch.zeros(R.shape[1], ld, dtype=R.dtype)
        Rt[:, :N] = R.t()
        return pre.apply_(Rt, torch.zeros_like(Rt))[:, :N].t().contiguous()

    t = probes.shape[1]
    Z = probes / probes.norm(dim=-2, keepdim=True)
    rhs = torch.cat([Z, y.unsqueeze(-1)], dim=-1) if owns_rhs else Z
    Khat = K + torch.diag(dvec)
    sol, Tm, info = OCG.linear_cg(lambda V: Khat @ V, rhs, n_tridiag=t, tolerance=1e-4, max_tridiag_iter=40, return_info=True,
                                  preconditioner=papply, mean_residual_fn=mean_fn)
    ld_part = OS.slq_logdet(Tm, N) * (t / t_total)
    return sol, ld_part, pre, info


def _kron_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from gpytorch_amd import distributed as D
    from gpytorch_amd import settings as S
    from gpytorch_amd.bbmm import deterministic_probe_matrix, structured_opts

    group = D.init_from_env("gloo")
    K, dvec, y = _kron_problem()
    N, t_total = K.shape[0], 6
    with S.sharding(probe_group=group), S.num_trace_samples(t_total), S.deterministic_probes(True):
        opts = structured_opts({}, torch.device("cpu"))          # the completion every structured autograd function performs
        assert opts["group"] is group and opts["t_total"] == t_total
        a, b = D.probe_shard(t_total, world, rank)
        assert opts["num_probes"] == b - a
        torch.manual_seed(99)                                     # identically seeded ranks draw the SAME t_total-column matrix ...
        probes = deterministic_probe_matrix(N, b - a, torch.device("cpu"), torch.float64, shard=(t_total, a, b))   # ... and keep their columns
        assert probes.shape == (N, b - a)

        def mean_fn(rnorm):
            return D.allreduce_residual_stats(rnorm.sum(), torch.tensor(float(rnorm.numel())), group)

        sol, ld_part, pre, info = _kron_solve(K, dvec, y, probes, t_total, rank == 0, mean_fn)
        D.allreduce_sum_(ld_part, group)
        ysol = sol[:, -1].clone() if rank == 0 else torch.zeros(N, dtype=sol.dtype)
        D.broadcast_(ysol, 0, group)
    q.put((rank, info["iters"], float(ld_part + pre.logdet), float(ysol @ y), probes.numpy(), (a, b)))
    dist.barrier()
    dist.destroy_process_group()


def test_probe_sharded_kronecker_mll_matches_single_process():
    """SURVEY.md 8e for the STRUCTURED operators: the Kronecker multitask system with its per-task (vector) noise, preconditioned
    by the row-callback pivoted Cholesky, probe-sharded over two ranks == the single-process evaluation with the same six probes
    (same iteration count, inverse quadratic form and log-determinant), and both agree with dense Cholesky."""
    sys.path.insert(0, ROOT)
    from gpytorch_amd import settings as S
    from gpytorch_amd.bbmm import deterministic_probe_matrix

    world, port = 2, free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_kron_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=180) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    K, dvec, y = _kron_problem()
    N, t_total = K.shape[0], 6
    with S.deterministic_probes(True):
        S.deterministic_probes.probe_vectors = None
        S.deterministic_probes._drawn.clear()
        torch.manual_seed(99)
        Z = deterministic_probe_matrix(N, t_total, torch.device("cpu"), torch.float64)
        sol, ld, pre, info = _kron_solve(K, dvec, y, Z, t_total, True)
        S.deterministic_probes.probe_vectors = None
        S.deterministic_probes._drawn.clear()
    logdet = float(ld + pre.logdet)
    inv_quad = float(sol[:, -1] @ y)
    for rank, iters, ldr, iqr, pr, (a, b) in results:
        assert torch.equal(torch.from_numpy(pr), Z[:, a:b])       # t_total DISTINCT probes over the ranks, the single-process ones
        assert iters == info["iters"]
        assert abs(ldr - logdet) < 1e-9 * abs(logdet)
        assert abs(iqr - inv_quad) < 1e-9 * abs(inv_quad)
    Khat = K + torch.diag(dvec)
    assert abs(inv_quad - float(y @ torch.linalg.solve(Khat, y))) < 1e-5 * abs(inv_quad)
    assert abs(logdet - float(torch.logdet(Khat))) < 0.1 * abs(float(torch.logdet(Khat)))       # six probes: a stochastic estimate


def _grid_worker(rank, world, port, q):
    """2-D split: rank = p * 2 + r -- probe group p owns a share of the probe columns, row half r a block of rows."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from gpytorch_amd import backend as B
    from gpytorch_amd import distributed as D
    from oracle import kernels as OK
    from oracle import linear_cg as OCG
    from oracle import slq as OS
    from tests.util import make_data

    D.init_from_env("gloo")
    G_r = 2
    p, r = divmod(rank, G_r)
    # every rank creates every subgroup, in the same order (torch.distributed contract)
    row_groups = [dist.new_group([pp * G_r + rr for rr in range(G_r)]) for pp in range(world // G_r)]
    probe_groups = [dist.new_group([pp * G_r + rr for pp in range(world // G_r)]) for rr in range(G_r)]
    row_group, probe_group = row_groups[p], probe_groups[r]
    n, t_total = 302, 6
    X, y = make_data(n, 3)
    Z = torch.randn(n, t_total, generator=torch.Generator().manual_seed(1234), dtype=torch.float64)
    Z = Z / Z.norm(dim=-2, keepdim=True)
    a, b = D.probe_shard(t_total, world // G_r, p)
    t = b - a
    rs = D.RowShard(B.PreparedPoints(X.clone(), n, 3, 3, "rbf"), row_group)
    assert rs.rank == r and rs.world == G_r
    K_rows = OK.kernel_matrix("rbf", X[rs.r0 : rs.r1], X, 0.25, 1.0, x1_eq_x2=False)
    cols = torch.cat([Z[:, a:b], y.unsqueeze(-1)], dim=-1) if p == 0 else Z[:, a:b]       # the y column rides with probe group 0
    rhs_loc = cols[rs.r0 : rs.r1]

    def mm_local(D_loc):      # this rank's rows of K_hat @ D: directions all-gathered over the ROW group only
        full = rs.gather(D_loc.t().contiguous())[:, :n].t()
        return K_rows @ full + 0.1 * D_loc

    def rowsum(v):            # inner products / norms: summed over the row group
        return D.allreduce_sum_(v.clone(), row_group)

    def mean_fn(rnorm):       # stopping rule: global mean over ALL columns -- summed over the probe group (one rank per probe share)
        return D.allreduce_residual_stats(rnorm.sum(), torch.tensor(float(rnorm.numel())), probe_group)

    sol_loc, T, info = OCG.linear_cg(mm_local, rhs_loc, n_tridiag=t, tolerance=1e-5, max_iter=300, max_tridiag_iter=30, return_info=True,
                                     rowsum_fn=rowsum, mean_residual_fn=mean_fn)
    ld = OS.slq_logdet(T, n) * (t / t_total)
    D.allreduce_sum_(ld, probe_group)
    q.put((rank, p, r, (a, b), (rs.r0, rs.r1), info["iters"], sol_loc.numpy(), float(ld)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_dimensional_split_probe_groups_times_row_halves_matches_single_process():
    """The 2-D split of DESIGN.md section 6 (probe groups x row halves; 4 ranks = 2 x 2 over gloo, float64): row-group collectives carry the
    products and inner products, the probe-group collectives the stopping rule and the SLQ sums -- two ORTHOGONAL subgroups per rank.
    Same iteration count, the same solves (rows x columns of every rank) and the same log-det as the single-process evaluation."""
    sys.path.insert(0, ROOT)
    from oracle import exact_gp as OG
    from oracle import linear_cg as OCG
    from oracle import slq as OS
    from tests.util import make_data

    world, port = 4, free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grid_worker, args=(rk, world, port, q)) for rk in range(world)]
    for pr in procs:
        pr.start()
    results = sorted([q.get(timeout=240) for _ in range(world)], key=lambda x: x[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    n, t_total = 302, 6
    X, y = make_data(n, 3)
    Z = torch.randn(n, t_total, generator=torch.Generator().manual_seed(1234), dtype=torch.float64)
    Z = Z / Z.norm(dim=-2, keepdim=True)
    rhs = torch.cat([Z, y.unsqueeze(-1)], dim=-1)
    sol, T, info = OCG.linear_cg(OG.make_matmul("rbf", X, 0.25, 1.0, 0.1), rhs, n_tridiag=t_total, tolerance=1e-5, max_iter=300, max_tridiag_iter=30,
                                 return_info=True)
    ld = float(OS.slq_logdet(T, n))
    scale = float(sol.abs().max())
    for rank, p, r, (a, b), (r0, r1), iters, s_loc, ldr in results:
        s_loc = torch.from_numpy(s_loc)
        assert iters == info["iters"], (rank, iters, info["iters"])
        # (row sharding changes the reduction ORDER: rounding differences grow along the recurrence up to the level of the CG tolerance, 1e-5)
        assert torch.allclose(s_loc[:, : b - a], sol[r0:r1, a:b], rtol=0, atol=3e-5 * scale)
        if p == 0:
            assert torch.allclose(s_loc[:, -1], sol[r0:r1, t_total], rtol=0, atol=3e-5 * scale)
        assert abs(ldr - ld) < 1e-6 * abs(ld), (rank, ldr, ld)


def test_layout_policy_cost_model():
    """distributed.choose_grid: the measured column ladder of the fused K*V (kernel generation replicated on every probe share) + one
    all-gather of the search directions per iteration over a row group.  The metric workload (64 probes) goes 1 x N -- probe sharding alone
    would leave 8 + 1 columns per GPU at N = 8: 45 ms against 91 / 8 + all-gather --, C4 (256 probes) keeps >= 2 probe shares, operators
    whose rows cannot be sharded stay probes-only, and nobody gets fewer than one probe or less than one 512-row group."""
    from gpytorch_amd import distributed as D

    # the ladder reproduces the measured launches (ms at n = 500 000): profiles/r05_s8_bench_kernel_stats.csv, DESIGN 3.2 / 6
    for cols, ms, c in ((1, 18.6, "split"), (2, 23.8, "split"), (9, 45.0, "split"), (33, 58.5, "split"), (65, 91.0, "split"), (65, 242.0, "f32"), (33, 131.0, "f32")):
        assert abs(D.kv_cost_ms(cols, 500_000, 500_000, c) - ms) < 0.02 * ms, (cols, c)
    for c in ("split", "f32"):
        assert [D.choose_grid(N, 500_000, 64, c) for N in (1, 2, 4, 8)] == [(1, 1), (1, 2), (1, 4), (1, 8)]
        P, R = D.choose_grid(8, 1_000_000, 256, c)
        assert P * R == 8 and P >= 2
    assert D.grid_cost_ms(1, 8, 500_000, 64) < 0.35 * D.grid_cost_ms(8, 1, 500_000, 64)          # 12.7 against 45 ms per iteration
    assert D.choose_grid(8, 500_000, 64, allow_rows=False) == (8, 1)
    assert D.choose_grid(8, 500_000, 4) == (1, 8)                                                  # fewer probes than ranks: rows only
    assert D.choose_grid(8, 2_000, 64)[1] <= 2                                                     # 2000 points: at most 3 row groups of 512
    P, R = D.choose_grid(6, 100_000, 10)
    assert P * R == 6


def _auto_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from gpytorch_amd import distributed as D
    from gpytorch_amd import settings as S

    D.init_from_env("gloo")
    out = {}
    assert S.sharding.mll_groups(500_000, 64) == (None, None) and not S.sharding.is_auto()
    with S.sharding("auto"):
        assert S.sharding.is_auto() and S.sharding.row_group() is dist.group.WORLD
        for name, (n, t) in (("metric", (500_000, 64)), ("c4", (1_000_000, 256)), ("probes_only", (3_000, 64))):
            pg, rg = S.sharding.mll_groups(n, t)
            a, b = torch.tensor([float(rank + 1)]), torch.tensor([float(rank + 1)])
            if pg is not None:
                dist.all_reduce(a, group=pg)
            if rg is not None:
                dist.all_reduce(b, group=rg)
            out[name] = (None if pg is None else dist.get_world_size(pg), None if rg is None else dist.get_world_size(rg), float(a), float(b))
            assert S.sharding.mll_groups(n, t) == (pg, rg)            # cached: a second evaluation creates no group
        pg, rg = S.sharding.mll_groups(500_000, 64, allow_rows=False)   # structured operators: probe columns only
        out["structured"] = (dist.get_world_size(pg), rg)
    assert not S.sharding.is_auto() and S.sharding.row_group() is None
    with pytest.raises(ValueError):
        S.sharding("auto", row_group=dist.group.WORLD)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_sharding_auto_builds_the_groups_of_the_chosen_grid():
    """settings.sharding("auto") on 4 gloo ranks: every MLL evaluation gets the subgroups of the grid the cost model picks for its (n, probes) --
    metric 1 x 4 (row group = WORLD, no probe group), C4 2 x 2 (rank = p * 2 + r), a small system probes-only -- built collectively on first
    use and cached."""
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_auto_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from gpytorch_amd.distributed import choose_grid

    assert choose_grid(4, 500_000, 64) == (1, 4) and choose_grid(4, 1_000_000, 256) == (2, 2) and choose_grid(4, 3_000, 64) == (4, 1)
    for r in range(4):
        assert got[r]["metric"] == (None, 4, float(r + 1), 10.0)
        p, rr = divmod(r, 2)
        assert got[r]["c4"] == (2, 2, float((rr + 1) + (2 + rr + 1)), float((2 * p + 1) + (2 * p + 2)))
        assert got[r]["probes_only"] == (4, None, 10.0, float(r + 1))
        assert got[r]["structured"] == (4, None)
