import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpytorch_amd import backend as B
from oracle import kernels as OK
dev = torch.device("cuda:0")
n = 60000
g = torch.Generator().manual_seed(0)
for cloud in ("normal", "uniform"):
    X = torch.randn(n, 3, generator=g) if cloud == "normal" else torch.rand(n, 3, generator=g)
    for kind, ls in (("matern52", 0.8), ("matern52", 3.0), ("matern32", 0.8), ("matern32", 3.0), ("rq", 0.5)):
        if cloud == "uniform":
            ls = ls / 4
        Xd = X.to(dev)
        par = 1.3 if kind == "rq" else None
        xp = B.prep_points(kind, Xd, torch.tensor([ls]), Xd.mean(0), par)
        mode = B.gram_mode(xp, xp)
        rows = torch.arange(0, n, 97)
        if kind == "rq":
            Kr = OK.rq(X[rows].double(), X.double(), ls, par, x1_eq_x2=False, direct=True)
        else:
            Kr = OK.kernel_matrix(kind, X[rows].double(), X.double(), ls, 1.0, x1_eq_x2=False, direct=True)
        for t in (1, 2, 3, 4):
            V = torch.randn(n, t, generator=torch.Generator().manual_seed(t))
            ref = Kr @ V.double()
            res = []
            for rep in range(2):
                out = B.kv(xp, xp, B.to_probe_major(V.to(dev)))[:, rows.to(dev)].t().double().cpu()
                err = (out - ref).abs().max(1).values / ref.abs().max()
                res.append((float(err.max()), int((err > 2e-5).sum())))
            print(cloud, kind, ls, "zmax2", round(xp.zmax2, 1), "mode", mode, "t", t, res, flush=True)
