"""Time the fused K*V at one shape (HIP events): python scripts/kv_time.py [n] [t] [reps] -> one JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, ".")
from gpytorch_amd import backend as B  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
t = int(sys.argv[2]) if len(sys.argv) > 2 else 65
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda:0")
torch.manual_seed(0)
X = torch.rand(n, 3, device=dev)
xp = B.prep_points("rbf", X, torch.tensor(0.25), X.mean(0))
V = torch.randn(t, B.round_up(n, 4), device=dev)
B.kv(xp, xp, V)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    B.kv(xp, xp, V)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(json.dumps(dict(n=n, t=t, ms=ms, tflops=2.0 * n * n * t / ms / 1e9, skew=os.environ.get("GPAMD_KV_SKEW", "0"))))
