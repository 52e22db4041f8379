import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpytorch_amd import backend as B
dev = torch.device("cuda:0")
n, d = 20000, 3
X = torch.rand(n, d, generator=torch.Generator().manual_seed(0)).to(dev)
xp = B.prep_points("rbf", X, torch.tensor(0.25), X.mean(0))
for t in (1, 8, 16, 17, 65):
    for name, mk in (("abs", lambda g: torch.randn(t, B.round_up(n, 4), generator=g).abs()), ("randn", lambda g: torch.randn(t, B.round_up(n, 4), generator=g))):
        g = torch.Generator().manual_seed(t)
        lt, rt = mk(g).to(dev), mk(g).to(dev)
        res = {}
        for split in (False, True):
            B.SPLIT_CONTRACTION = split
            res[split] = B.kv_grad2(xp, xp, lt, rt, iso=False)[0].double().cpu()
        B.SPLIT_CONTRACTION = None
        a, b = res[False], res[True]
        print(t, name, "fp32", [f"{v:.6e}" for v in a[:5].tolist()], "split", [f"{v:.6e}" for v in b[:5].tolist()], "rel", [f"{v:.1e}" for v in ((a - b).abs() / a.abs().clamp_min(1e-30))[:5].tolist()], flush=True)
