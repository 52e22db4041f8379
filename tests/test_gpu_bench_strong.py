"""The driver's multi-GPU command line, end to end on one device: `python bench.py --gpus N` (default: STRONG scaling of the metric workload on the
layout `gpytorch_amd.distributed.choose_grid` picks for the nominal size -- 1 x N: every rank keeps the 64 + 1 columns on 1 / N of the rows)
reproduces the single-process evaluation.  N gloo ranks share cuda:0 (GPAMD_BENCH_BACKEND=gloo GPAMD_BENCH_SHARE_DEVICE=1: the same code path as
RCCL apart from the transport -- two RCCL ranks cannot share a device); `--size 4096` keeps it to seconds.  Replaces the layout a user of
`gpytorch/kernels/multi_device_kernel.py:24-92` never had to choose."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(n_ranks, *extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(GPAMD_BENCH_BACKEND="gloo", GPAMD_BENCH_SHARE_DEVICE="1", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n_ranks), "--steps", "1", "--warmup", "0", "--size", "4096",
                          "--skip-split", "--skip-parity", "--skip-cpu-baseline", "--skip-extras", *extra], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("contraction", ["f32", "split"])
def test_strong_scaling_line_reproduces_the_single_process_step(contraction, dev):
    one = _bench(1, "--contraction", contraction)
    four = _bench(4, "--contraction", contraction)
    assert one["scaling"] == four["scaling"] == "strong" and one["config"]["grid"] == [1, 1] and four["config"]["grid"] == [1, 4]
    assert four["n_gpus"] == 4 and four["config"]["probes_total"] == one["config"]["probes_total"] == 64 and four["config"]["rhs_columns_rank0"] == 65
    # same probes (one probe share: every rank draws from the seed of share 0), same stopping iteration, same value
    assert abs(four["config"]["cg_iterations_per_step"] - one["config"]["cg_iterations_per_step"]) <= 1
    # (row blocks run rectangular launches: other split counts and summation orders than the square product.  At cg_tolerance 1 on 4096 points the
    # value is a LOOSE solve's -- it moves by 1.6e-3 under the rounding of the split contraction (2e-5 per product), by < 5e-4 under the fp32 MFMAs)
    assert abs(four["mll"] - one["mll"]) < (5e-4 if contraction == "f32" else 5e-3) * max(1.0, abs(one["mll"])), (four["mll"], one["mll"])
    # the line's arithmetic: whole-job flops = 2 n^2 (64 + 1) x iterations whatever the layout
    for rec in (one, four):
        its = rec["config"]["cg_iterations_per_step"]
        assert abs(rec["value"] - 2.0 * 4096**2 * 65 * its / (rec["ms_per_step"] * 1e-3) / 1e12) < 1e-6 * rec["value"] + 1e-12


def test_weak_scaling_line_is_kept_behind_a_flag(dev):
    two = _bench(2, "--scaling", "weak")
    assert two["scaling"] == "weak" and two["config"]["grid"] == [2, 1] and two["config"]["probes_total"] == 128 and two["config"]["probes_rank0"] == 64
