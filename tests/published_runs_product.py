"""The two notebook trainings of ``tests/golden/reference_notebook_runs.npz`` written against THIS repository's gpytorch-shaped API, as a user of
the reference would write them (model classes as in the notebooks' cells; only the import changes).  Shared by the CPU wiring test
(``tests/test_published_runs_cpu.py``, native entry points doubled by ``tests/shim/cpu_backend.py``) and the device test
(``tests/test_gpu_published_runs.py``, the HIP library)."""
import os

import numpy as np
import torch

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_notebook_runs.npz"))


def _t(name, dev):
    return torch.from_numpy(G[name]).to(dev)


def hadamard_run(g, dev, k, data, steps=100):
    """examples/03_Multitask_Exact_GPs/Hadamard_Multitask_GP_Regression.ipynb, ``train_model`` with ``GaussianLikelihood`` (the shared-noise runs:
    model 0 on the 40 points, model 2 on the 2 x 10-point subset).  Returns (loss before every step, the learned noise after the last)."""
    x, i, y = _t(f"{data}_x", dev), _t(f"{data}_i", dev).unsqueeze(-1), _t(f"{data}_y", dev)

    class MultitaskGPModel(g.models.ExactGP):
        def __init__(self, train_x, train_y, likelihood):
            super().__init__(train_x, train_y, likelihood)
            self.mean_module = g.means.ConstantMean()
            self.covar_module = g.kernels.RBFKernel()
            self.task_covar_module = g.kernels.IndexKernel(num_tasks=2, rank=1)

        def forward(self, x, i):
            covar = self.covar_module(x).mul(self.task_covar_module(i))
            return g.distributions.MultivariateNormal(self.mean_module(x), covar)

    likelihood = g.likelihoods.GaussianLikelihood().to(dev)
    model = MultitaskGPModel((x, i), y, likelihood).to(dev)
    with torch.no_grad():      # the draws the reference's IndexKernel constructor made in the authors' run (index_kernel.py:69-72)
        model.task_covar_module.covar_factor.copy_(_t(f"had_init_covar_factor_{k}", dev))
        model.task_covar_module.raw_var.copy_(_t(f"had_init_raw_var_{k}", dev))
    model.train()
    likelihood.train()
    optimizer = torch.optim.Adam(model.parameters(), lr=0.1)
    mll = g.mlls.ExactMarginalLogLikelihood(likelihood, model)
    losses = []
    for _ in range(steps):
        optimizer.zero_grad()
        output = model(x, i)
        loss = -mll(output, y)
        loss.backward()
        losses.append(float(loss.detach()))
        optimizer.step()
    return losses, float(likelihood.noise.detach().reshape(-1)[0])


def classification_run(g, dev, steps=46):
    """examples/01_Exact_GPs/GP_Regression_on_Classification_Labels.ipynb, cell for cell: ``DirichletClassificationLikelihood`` turns the labels into
    regression targets + fixed per-point noise for a batch of three exact GPs (one per class) and learns one more noise per member.  Returns rows of
    (loss, mean lengthscale, mean learned noise) as the notebook prints them, one per step."""
    train_x, train_y = _t("cls_x", dev), _t("cls_labels", dev)
    likelihood = g.likelihoods.DirichletClassificationLikelihood(train_y, learn_additional_noise=True).to(dev)
    # the class's own transform against the reference's (the fixture holds the output of the reference's ``_prepare_targets``, executed)
    assert torch.allclose(likelihood.transformed_targets, _t("cls_targets", dev), rtol=1e-6, atol=1e-6)
    assert torch.allclose(likelihood.noise_covar.noise, _t("cls_fixed_noise", dev), rtol=1e-6, atol=1e-6)
    X, targets = train_x, likelihood.transformed_targets
    bs = torch.Size((likelihood.num_classes,))

    class DirichletGPModel(g.models.ExactGP):
        def __init__(self, train_x, train_y, likelihood):
            super().__init__(train_x, train_y, likelihood)
            self.mean_module = g.means.ConstantMean(batch_shape=bs)
            self.covar_module = g.kernels.ScaleKernel(g.kernels.RBFKernel(batch_shape=bs), batch_shape=bs)

        def forward(self, x):
            return g.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    model = DirichletGPModel(X, targets, likelihood).to(dev)
    model.train()
    likelihood.train()
    optimizer = torch.optim.Adam(model.parameters(), lr=0.1)
    mll = g.mlls.ExactMarginalLogLikelihood(likelihood, model)
    rows = []
    for _ in range(steps):
        optimizer.zero_grad()
        output = model(X)
        loss = -mll(output, targets).sum()
        loss.backward()
        rows.append((float(loss.detach()), float(model.covar_module.base_kernel.lengthscale.detach().mean()), float(likelihood.second_noise_covar.noise.detach().mean())))
        optimizer.step()
    return rows
