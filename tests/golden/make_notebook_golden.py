#!/usr/bin/env python3
"""Golden fixtures from the reference's own PUBLISHED RUNS: two example notebooks under ``/root/reference/examples`` fix their random seed, build
their data from it and carry the text their authors' run printed -- losses, lengthscales and noises along an Adam trajectory of
``ExactMarginalLogLikelihood`` -- produced by the real ``gpytorch`` on the real ``linear_operator`` (the package this image cannot install: SURVEY.md 8c).
These are the only outputs of the complete reference stack that exist here, and they are reproducible: ``torch.manual_seed`` + the CPU generator
give the same draws today.

Run once in the build container (``python tests/golden/make_notebook_golden.py``); ``reference_notebook_runs.npz`` is committed and travels to the
GPU box (``/root/reference`` does not exist there).  What is EXECUTED from the reference (nothing is copied into the repository):

  * ``examples/03_Multitask_Exact_GPs/Hadamard_Multitask_GP_Regression.ipynb``: the data cell (``torch.manual_seed(1)``, two tasks of 20 points,
    noises sqrt(0.3) / sqrt(0.1)) is exec'd as it stands; the printed ``Iter k/100 - Loss`` lines of its four trainings (shared noise and per-task
    noise, 40 points and the 2 x 10-point subset) and the two printed ``likelihood.noise`` tensors are parsed from the output cells.  The initial
    ``IndexKernel`` parameters are what the global generator yields next, in the order the kernel's constructor draws them
    (``gpytorch/kernels/index_kernel.py:69-72``: ``randn(num_tasks, rank)`` then ``randn(num_tasks)``), once per model the notebook builds --
    nothing else in the notebook consumes the generator (its predictions at n = 40 take the Cholesky route);
  * ``examples/01_Exact_GPs/GP_Regression_on_Classification_Labels.ipynb``: ``gen_data`` (seed 2019, 500 points, d = 2, three classes) is exec'd as
    it stands; the Dirichlet transform of the labels is the reference's own ``DirichletClassificationLikelihood._prepare_targets``
    (``gpytorch/likelihoods/gaussian_likelihood.py:397-413``), extracted with ``ast`` and executed; the ten printed
    ``Iter k/50 - Loss / lengthscale / noise`` lines are parsed from the output cell.  The model: a batch of three exact GPs (ConstantMean,
    ScaleKernel(RBFKernel), per-point fixed noise + one learned noise per member), loss = minus the SUM of the three per-datum MLLs.
"""
from __future__ import annotations

import json
import math
import os
import re
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import _extract_method  # noqa: E402

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _cells(path):
    nb = json.load(open(path))
    return [("".join(c["source"]), "".join("".join(o.get("text", [])) for o in c.get("outputs", []))) for c in nb["cells"] if c["cell_type"] == "code"]


def hadamard_notebook():
    path = f"{REF}/examples/03_Multitask_Exact_GPs/Hadamard_Multitask_GP_Regression.ipynb"
    cells = _cells(path)
    (data_src,) = [s for s, _ in cells if "torch.manual_seed(1)" in s]
    ns = {"torch": torch, "math": math}
    exec(compile(data_src, path, "exec"), ns)                  # the notebook's own data cell: seeds the generator, draws inputs and noise
    inits = [(torch.randn(2, 1), torch.randn(2)) for _ in range(4)]   # index_kernel.py:69-72, one pair per model built, in the notebook's order
    outs = [o for s, o in cells if "train_model(" in s and "Iter" in o]
    assert len(outs) == 3, len(outs)
    loss = [[float(v) for v in re.findall(r"Iter \d+/100 - Loss: ([0-9.]+)", o)] for o in outs]
    runs = [loss[0], loss[1], loss[2][:4], loss[2][4:]]        # shared noise; per-task noise; subset shared; subset per-task
    assert all(len(r) == 4 for r in runs), runs
    noises = [[float(v) for v in m.split(",")] for m in re.findall(r"likelihood\.noise=tensor\(\[([0-9., ]+)\]", outs[2])]
    assert len(noises) == 2 and len(noises[0]) == 1 and len(noises[1]) == 2, noises
    out = {
        "had_x": ns["full_train_x"].numpy(), "had_i": ns["full_train_i"].squeeze(-1).numpy(), "had_y": ns["full_train_y"].numpy(),
        "had_printed_loss": np.array(runs), "had_printed_iterations": np.array([25, 50, 75, 100]),
        "had_final_noise_subset_shared": np.array(noises[0]), "had_final_noise_subset_per_task": np.array(noises[1]),
    }
    # the last cell's subset: 10 points per task, BOTH tasks at the first task's noise level (the notebook's TASK_NOISE)
    N, tn = 10, ns["TASK_NOISES"][0]
    out["had_sub_x"] = torch.cat([ns["train_x1"][:N], ns["train_x2"][:N]]).numpy()
    out["had_sub_i"] = torch.cat([ns["train_i_task1"][:N], ns["train_i_task2"][:N]]).squeeze(-1).numpy()
    out["had_sub_y"] = (torch.cat([ns["train_f1"][:N], ns["train_f2"][:N]]) + torch.cat([tn * ns["train_noise1"][:N], tn * ns["train_noise2"][:N]])).numpy()
    for k, (cf, rv) in enumerate(inits):
        out[f"had_init_covar_factor_{k}"] = cf.numpy()
        out[f"had_init_raw_var_{k}"] = rv.numpy()
    return out


def classification_notebook():
    path = f"{REF}/examples/01_Exact_GPs/GP_Regression_on_Classification_Labels.ipynb"
    cells = _cells(path)
    (gen_src,) = [s for s, _ in cells if "def gen_data" in s]
    ns = {"torch": torch}
    exec(compile(gen_src, path, "exec"), ns)
    train_x, train_y, _ = ns["gen_data"](500)
    prep = _extract_method(f"{REF}/gpytorch/likelihoods/gaussian_likelihood.py", "DirichletClassificationLikelihood", "_prepare_targets", {"torch": torch})
    sigma2, transformed, num_classes = prep(None, train_y)      # ([C, n] fixed noise, [n, C] regression targets, C)
    (printed,) = [o for s, o in cells if "training_iter" in s and "Iter" in o]
    rows = re.findall(r"Iter (\d+)/50 - Loss: ([0-9.]+)\s+lengthscale: ([0-9.]+)\s+noise: ([0-9.]+)", printed)
    assert len(rows) == 10 and num_classes == 3, (len(rows), num_classes)
    return {
        "cls_x": train_x.numpy(), "cls_labels": train_y.numpy(), "cls_fixed_noise": sigma2.numpy(), "cls_targets": transformed.t().contiguous().numpy(),
        "cls_printed_iterations": np.array([int(r[0]) for r in rows]),
        "cls_printed_loss_lengthscale_noise": np.array([[float(v) for v in r[1:]] for r in rows]),
    }


if __name__ == "__main__":
    out = {}
    out.update(hadamard_notebook())
    out.update(classification_notebook())
    np.savez(os.path.join(OUT, "reference_notebook_runs.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, v.dtype)
