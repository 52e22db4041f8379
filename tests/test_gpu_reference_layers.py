"""Device half of tests/test_reference_layers_cpu.py: the reference's own ExactGP / GaussianLikelihood / ExactMarginalLogLikelihood /
DefaultPredictionStrategy over the plugin operator on the BBMM branches (mBCG + SLQ forward, fused bilinear-derivative backward, mean-cache CG
and LOVE Lanczos), n = 3000 > max_cholesky_size, against the standalone layers with the same probe vectors.

Needs BOTH a ROCm device and the reference checkout: the build container has the checkout and no GPU, the GPU box has no checkout (the
reference cannot travel) -- so this test is SKIPPED in both today and documents the run for a machine that has the two; the CPU half is what
executes in the build container."""
import os

import pytest
import torch

from tests.test_reference_layers_cpu import REF, _run

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "gpytorch")), reason="the reference checkout is not present on this machine")]


@pytest.fixture()
def reference_on_device(monkeypatch):
    import importlib
    import sys

    from tests.test_reference_layers_cpu import SHIM

    monkeypatch.syspath_prepend(REF)
    monkeypatch.syspath_prepend(SHIM)
    before = set(sys.modules)
    lo = importlib.import_module("linear_operator")
    gp = importlib.import_module("gpytorch")
    from gpytorch_amd import dropin

    yield gp, lo, dropin.build(gp, lo)
    for name in set(sys.modules) - before:
        if name.split(".")[0] in ("gpytorch", "linear_operator"):
            del sys.modules[name]


def test_reference_layers_on_the_bbmm_branches(reference_on_device, dev):
    gp, lo, ns = reference_on_device
    import gpytorch_amd as g

    n, d = 3000, 3
    gen = torch.Generator().manual_seed(0)
    X = torch.rand(n, d, generator=gen).to(dev)
    y = (torch.sin(6 * X[:, 0].cpu()) + 0.1 * torch.randn(n, generator=gen)).to(dev)
    Xs = torch.rand(50, d, generator=gen).to(dev)
    hp = (0.3, 1.4, 0.05, 0.2)
    S = g.settings                     # (the shim's settings ARE these classes: one shared state for both stacks)
    S.deterministic_probes.probe_vectors = torch.randn(n, 16, generator=gen).to(dev)
    try:
        with S.max_cholesky_size(0), S.deterministic_probes(True), S.cg_tolerance(1e-3), S.eval_cg_tolerance(1e-4), S.num_trace_samples(16):
            ref = _run(gp, lambda: ns.RBFKernel().to(dev), X, y, Xs, hp, lambda: gp.settings.lazily_evaluate_kernels(False))
            own = _run(g, lambda: g.kernels.RBFKernel().to(dev), X, y, Xs, hp, lambda: g.settings.lazily_evaluate_kernels(False))
    finally:
        S.deterministic_probes.probe_vectors = None
    assert abs(ref[0] - own[0]) < 1e-6 * max(1.0, abs(own[0]))
    for k in ref[1]:
        assert torch.allclose(ref[1][k], own[1][k], rtol=1e-4, atol=1e-6), k
    assert torch.allclose(ref[2], own[2], rtol=1e-4, atol=1e-5) and torch.allclose(ref[3], own[3], rtol=1e-3, atol=1e-5)
