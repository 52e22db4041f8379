"""bench.py's self-launch: `python bench.py --gpus N` WITHOUT a launcher in the environment (the shape of the driver's N = 1 command) must start N
ranks itself under torch.distributed.run on the loopback address and keep the one-JSON-line contract on rank 0.  No GPU needed: the
GPAMD_BENCH_LAUNCH_ONLY hook stops every rank after one gloo all-reduce, before anything touches a device."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, extra=()):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(GPAMD_BENCH_LAUNCH_ONLY="1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", *extra], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout            # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks_when_no_launcher_is_present():
    """... and the default multi-GPU line is STRONG scaling of the metric workload (64 probes in total) on the layout the cost model picks for the
    nominal size: every rank keeps all the columns on a third of the rows."""
    rec = _run(3)
    assert rec["launched"] == 3 and rec["rank_sum"] == 6.0 and rec["n_gpus"] == 3
    assert rec["scaling"] == "strong" and rec["grid"] == [1, 3]
    assert [list(x) for x in rec["grid_sums"]] == [[0, 1.0, 6.0], [1, 2.0, 6.0], [2, 3.0, 6.0]]     # row group = all three ranks, no probe group


def test_bench_single_rank_needs_no_launcher():
    rec = _run(1)
    assert rec["launched"] == 1 and rec["grid"] == [1, 1]


def test_metric_smoke_run_takes_the_layout_of_the_full_configuration():
    """`--gpus 4 --config metric --size 4096`: the layout is chosen for the configuration's NOMINAL size (n = 500 000, 64 probes -> 1 x 4), so
    that a shrunken smoke run exercises the collectives of the real one; `--scaling weak` keeps rounds 1-5's probes-only line."""
    rec = _run(4, ("--config", "metric", "--size", "4096"))
    assert rec["grid"] == [1, 4] and rec["scaling"] == "strong"
    rec = _run(4, ("--config", "metric", "--size", "4096", "--scaling", "weak"))
    assert rec["grid"] == [4, 1] and rec["scaling"] == "weak"
    rec = _run(8, ("--config", "c4"))
    assert rec["grid"] == [2, 4] and rec["scaling"] == "strong"


def test_grid_subgroups_of_the_two_dimensional_split():
    """`--grid 2x2` on 4 ranks (rank = p * 2 + r): the probe group of a rank is {r, 2 + r}, its row group {2 p, 2 p + 1} -- checked by
    all-reducing (rank + 1) inside each subgroup over gloo."""
    rec = _run(4, ("--config", "c4", "--grid", "2x2"))
    assert rec["launched"] == 4
    want = [[0, 1.0 + 3.0, 1.0 + 2.0], [1, 2.0 + 4.0, 1.0 + 2.0], [2, 1.0 + 3.0, 3.0 + 4.0], [3, 2.0 + 4.0, 3.0 + 4.0]]
    assert [list(x) for x in rec["grid_sums"]] == want
