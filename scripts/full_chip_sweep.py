import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpytorch_amd import backend as B
from oracle import kernels as OK
dev = torch.device("cuda:0")
n = 60000
kinds = sys.argv[1].split("+") if len(sys.argv) > 1 else ["matern52", "matern32", "rbf", "rq"]
dims = [int(v) for v in sys.argv[2].split("+")] if len(sys.argv) > 2 else [1, 3]
for kind in kinds:
    for d in dims:
        X = torch.rand(n, d, generator=torch.Generator().manual_seed(d))
        ls = {1: 0.25, 2: 0.4, 3: 0.6, 6: 1.0, 10: 1.4}[d]
        par = 1.3 if kind == "rq" else None
        rows = torch.arange(0, n, 127)
        Kr = OK.rq(X[rows].double(), X.double(), ls, par, x1_eq_x2=False, direct=True) if kind == "rq" else \
            OK.kernel_matrix(kind, X[rows].double(), X.double(), ls, 1.0, x1_eq_x2=False, direct=True)
        xp = B.prep_points(kind, X.to(dev), torch.tensor([ls]), X.mean(0).to(dev), par)
        for split in (True, False):
            B.SPLIT_CONTRACTION = split
            bad = []
            for t in list(range(1, 35)) + [48, 64, 65, 66, 97, 129]:
                V = torch.randn(n, t, generator=torch.Generator().manual_seed(t))
                ref = Kr @ V.double()
                worst = 0.0
                for rep in range(3):
                    out = B.kv(xp, xp, B.to_probe_major(V.to(dev)))[:, rows.to(dev)].t().double().cpu()
                    worst = max(worst, float((out - ref).abs().max() / ref.abs().max()))
                if worst > 2e-5:
                    bad.append((t, float(f"{worst:.2e}")))
            print(kind, "d", d, "split" if split else "f32", "bad:", bad, flush=True)
B.SPLIT_CONTRACTION = None
