#!/usr/bin/env python3
"""Launch the product fused K*V kernel a few times at the bench shape (for rocprofv3 --pmc passes)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpytorch_amd import backend as B  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
t = int(sys.argv[2]) if len(sys.argv) > 2 else 65
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda:0")
torch.manual_seed(0)
X = torch.rand(n, 3, device=dev)
xp = B.prep_points("rbf", X, torch.tensor(0.25), X.mean(0))  # centred, like RBFKernel.forward -> Gram-form kernel
V = torch.randn(t, B.round_up(n, 4), device=dev)
for _ in range(reps):
    out = B.kv(xp, xp, V)
torch.cuda.synchronize()
print("ok", float(out.abs().max()))
