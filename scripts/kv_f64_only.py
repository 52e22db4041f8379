#!/usr/bin/env python3
"""Launch the fused float64 K*V kernel (csrc/kv_f64.hpp) a few times (for rocprofv3 passes): python scripts/kv_f64_only.py n d t reps"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpytorch_amd import backend as B  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t = int(sys.argv[3]) if len(sys.argv) > 3 else 65
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0")
torch.manual_seed(0)
X = torch.rand(n, d, device=dev, dtype=torch.float64)
xp = B.prep_points("rbf", X, torch.tensor(0.25 if d <= 3 else 0.8, dtype=torch.float64), X.mean(0))
assert B.fused_f64(xp, xp)
V = torch.randn(t, B.round_up(n, 4), device=dev, dtype=torch.float64)
for _ in range(reps):
    out = B.kv(xp, xp, V)
torch.cuda.synchronize()
print("ok", float(out.abs().max()))
