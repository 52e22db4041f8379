"""The oracle against the reference's own PUBLISHED outputs (``tests/golden/reference_notebook_runs.npz``: text printed by the authors' runs of two
example notebooks on the real ``gpytorch`` + ``linear_operator`` stack, and the seed-determined data behind them -- ``make_notebook_golden.py``).

Printed to three decimals: a value must round to the printed one, i.e. lie within 5e-4 of it (+ 1e-4 for the float32 arithmetic of the authors'
run against the float64 oracle).  50-100 Adam steps sit between the first and the last printed number, so the GRADIENTS are pinned with the values."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import published_runs as PR

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_notebook_runs.npz"))
TOL = 6e-4
pytestmark = pytest.mark.filterwarnings("ignore:CG terminated", "ignore:Converting a tensor")


def _t(name, dtype=torch.float64):
    a = torch.from_numpy(G[name])
    return a.to(dtype) if a.is_floating_point() else a


HADAMARD_RUNS = [  # (model index in the notebook's order, data, per-task noise)
    (0, "had", False),
    (1, "had", True),
    (2, "had_sub", False),
    (3, "had_sub", True),
]


@pytest.mark.parametrize("k,data,per_task", HADAMARD_RUNS)
def test_hadamard_multitask_notebook_losses(k, data, per_task):
    """examples/03_Multitask_Exact_GPs/Hadamard_Multitask_GP_Regression.ipynb: ``Iter 25 / 50 / 75 / 100 - Loss`` of its four trainings, and the
    learned noises it prints after the two subset runs."""
    x, i, y = _t(f"{data}_x"), _t(f"{data}_i"), _t(f"{data}_y")
    p = PR.hadamard_parameters(G[f"had_init_covar_factor_{k}"], G[f"had_init_raw_var_{k}"], per_task)
    losses, snaps = PR.adam_trajectory(lambda q: PR.hadamard_loss(q, x, i, y), p, 100, snapshot_at=(25, 100))
    got = [losses[it - 1] for it in G["had_printed_iterations"]]
    assert np.abs(np.array(got) - G["had_printed_loss"][k]).max() < TOL, (got, G["had_printed_loss"][k])
    noise = (F.softplus(p["raw_noise"]) + PR.NOISE_FLOOR).detach().numpy()   # after the 100th step: what the notebook prints next
    if k == 2:
        assert np.abs(noise - G["had_final_noise_subset_shared"]).max() < 1e-4, noise
    if k == 3:
        assert np.abs(noise - G["had_final_noise_subset_per_task"]).max() < 1e-4, noise
    # the BBMM restatement (mBCG + Lanczos quadrature, unit-vector probes = exact trace) prints the same numbers at the same parameters
    for it, q in snaps.items():
        with torch.no_grad():
            khat, r = PR.hadamard_khat_and_residual(q, x, i, y)
            via_bbmm = float(PR.bbmm_loss(khat, r))
        assert abs(via_bbmm - losses[it - 1]) < 1e-6, (it, via_bbmm, losses[it - 1])
        assert abs(via_bbmm - G["had_printed_loss"][k][list(G["had_printed_iterations"]).index(it)]) < TOL


def test_classification_labels_notebook_trajectory():
    """examples/01_Exact_GPs/GP_Regression_on_Classification_Labels.ipynb: ten printed (loss, mean lengthscale, mean learned noise) triples of a
    batch of three exact GPs with fixed per-point noise + a learned noise per member."""
    X, targets, fixed = _t("cls_x"), _t("cls_targets"), _t("cls_fixed_noise")
    p = PR.classification_parameters(3)
    seen = []

    def loss_fn(q):
        seen.append((float(F.softplus(q["raw_lengthscale"]).mean()), float((F.softplus(q["raw_noise"]) + PR.NOISE_FLOOR).mean())))
        return PR.classification_loss(q, X, targets, fixed)

    losses, snaps = PR.adam_trajectory(loss_fn, p, 46, snapshot_at=(1, 46))
    its = G["cls_printed_iterations"]
    got = np.array([[losses[it - 1], *seen[it - 1]] for it in its])
    assert np.abs(got - G["cls_printed_loss_lengthscale_noise"]).max() < TOL, (got, G["cls_printed_loss_lengthscale_noise"])
    for it, q in snaps.items():
        with torch.no_grad():
            khat, r = PR.classification_khat_and_residual(q, X, targets, fixed)
            via_bbmm = float(PR.bbmm_loss(khat, r))
        assert abs(via_bbmm - losses[it - 1]) < 1e-6, (it, via_bbmm, losses[it - 1])


def test_product_layers_reproduce_the_published_runs_on_the_cpu_double(monkeypatch):
    """The same trainings written against ``gpytorch_amd``'s API (``tests/published_runs_product.py``), with the native entry points of the
    small-n branches doubled on the CPU (``tests/shim/cpu_backend.py``): the HOST layers -- ExactGP, IndexKernel and the Hadamard operator, batch
    members with fixed + per-member learned noise, ExactMarginalLogLikelihood and every gradient -- land on the printed numbers.  The device run
    of the same file (the HIP library instead of the double, and the BBMM branches) is ``tests/test_gpu_published_runs.py``."""
    from tests.shim import cpu_backend

    cpu_backend.install(monkeypatch)
    import gpytorch_amd as g
    from tests import published_runs_product as P

    dev = torch.device("cpu")
    for k, data in ((0, "had"), (2, "had_sub")):
        losses, noise = P.hadamard_run(g, dev, k, data)
        got = np.array([losses[it - 1] for it in G["had_printed_iterations"]])
        assert np.abs(got - G["had_printed_loss"][k]).max() < TOL + 1e-4, (got, G["had_printed_loss"][k])
        if k == 2:
            assert abs(noise - float(G["had_final_noise_subset_shared"][0])) < 2e-4, noise
    rows = np.array(P.classification_run(g, dev, 21))
    want = G["cls_printed_loss_lengthscale_noise"][:5]
    assert np.abs(rows[G["cls_printed_iterations"][:5] - 1] - want).max() < TOL + 1e-4, (rows, want)
