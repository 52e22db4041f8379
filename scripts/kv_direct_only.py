#!/usr/bin/env python3
"""Launch the direct-difference + split-contraction kernel (csrc/kv_directh.hpp) a few times at the road3d-like shape (for rocprofv3 --pmc passes):
python scripts/kv_direct_only.py [n] [t] [reps] [kind]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpytorch_amd import backend as B  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 217_437
t = int(sys.argv[2]) if len(sys.argv) > 2 else 11
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
kind = sys.argv[4] if len(sys.argv) > 4 else "matern52"
dev = torch.device("cuda:0")
torch.manual_seed(0)
X = torch.rand(n, 3, device=dev)
xp = B.prep_points(kind, X, torch.tensor(0.05), X.mean(0))
V = torch.randn(t, B.round_up(n, 4), device=dev)
B.FORCE_KV_FLAGS = B.KV_SPLIT
for _ in range(reps):
    out = B.kv(xp, xp, V)
torch.cuda.synchronize()
print("ok", float(out.abs().max()))
