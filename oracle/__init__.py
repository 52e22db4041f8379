"""CPU oracle for the BBMM ExactGP hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (CPU, float64 by default) restatement of the
reference algorithms on the hot path named by BASELINE.json:north_star.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it; the product package ``gpytorch_amd`` never does.

Pinning status
--------------
* Kernel arithmetic (RBF / Matern / sq_dist / dist): PINNED.  ``tests/golden/
  make_golden.py`` executes the reference's own ``functions/rbf_covariance.py``,
  ``functions/matern_covariance.py`` and ``kernels/kernel.py::sq_dist/dist`` in
  this container and stores their outputs in ``tests/golden/*.npz``;
  ``tests/test_oracle_golden.py`` checks this oracle against those files and
  against the hand-computed known answers of the reference's unit tests.
* Parameter transforms (softplus / sigmoid inverses behind every constrained
  hyper-parameter): PINNED the same way (``gpytorch/utils/transforms.py`` executed,
  ``tests/golden/transform_values.npz``).
* MVN log-prob / MLL assembly / predictive equations: PINNED to dense float64
  Cholesky (the same deterministic ground truth every reference test compares
  against) and to the known answer -4.8157 of
  ``test/distributions/test_multivariate_normal.py:40-43``.
* The WHOLE exact-GP objective and its gradients: PINNED to outputs of the complete reference stack (``gpytorch`` on the real
  ``linear_operator``) -- the text the authors' runs of two seed-fixed example notebooks printed and the reference ships
  (``examples/03_Multitask_Exact_GPs/Hadamard_Multitask_GP_Regression.ipynb``: four 100-step Adam trainings of a Hadamard multitask
  ExactGP, shared and per-task noise; ``examples/01_Exact_GPs/GP_Regression_on_Classification_Labels.ipynb``: 46 steps of a batch of
  three exact GPs with fixed + learned noise).  ``tests/golden/make_notebook_golden.py`` executes the notebooks' data cells and parses
  their output cells into ``tests/golden/reference_notebook_runs.npz``; ``oracle/published_runs.py`` lands on every printed digit
  (46 numbers: losses, lengthscales, noises), by dense Cholesky AND through this package's mBCG + Lanczos-quadrature restatement with
  unit-vector probes (``tests/test_published_runs_cpu.py``).
* mBCG / Lanczos / pivoted Cholesky / preconditioner / SLQ (the arithmetic of
  the third-party ``linear_operator>=0.6.1`` package, which is NOT vendored in
  /root/reference and is not installable here): restated from the published
  algorithm.  Iteration-level parity is UNPINNED -- the
  reference holds no golden vectors for CG iterates, tridiagonal matrices or
  pivoted-Cholesky factors; these routines are pinned only through their
  results: against dense Cholesky (solve, log-det given probes, predictive
  mean/variance) within the tolerances the reference's own tests use, and --
  for the value mBCG + SLQ return -- against the published notebook losses above.
"""
from . import kernels, linear_cg, pivoted_cholesky, lanczos, slq, exact_gp, multitask, published_runs  # noqa: F401
