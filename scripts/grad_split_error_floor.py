import torch, sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpytorch_amd import backend as B
dev = torch.device("cuda:0")
def run(n, d, t, mode, ls=0.6):
    gen = torch.Generator().manual_seed(0)
    x = torch.rand(n, d, generator=gen)
    L = torch.randn(n, t, generator=gen); R = torch.randn(n, t, generator=gen)
    if mode == "abs": L, R = L.abs(), R.abs()
    if mode == "scaled": L = L * torch.logspace(-3, 3, t).unsqueeze(0); R = R * torch.logspace(2, -2, t).unsqueeze(0)
    xp = B.prep_points("rbf", x.to(dev), torch.tensor([ls]), x.mean(0).to(dev))
    z = xp.xp[:, :d].double().cpu()
    Kz = torch.exp2(-(z.unsqueeze(1) - z.unsqueeze(0)).pow(2).sum(-1))
    W = L.double() @ R.double().t()
    true0 = float((W * Kz).sum()); scale = float((W.abs() * Kz).sum())
    lt, rt = B.to_probe_major(L.to(dev)), B.to_probe_major(R.to(dev))
    out = []
    for split in (False, True):
        B.SPLIT_CONTRACTION = split
        g, _ = B.kv_grad2(xp, xp, lt, rt, iso=True)
        out.append(abs(float(g[0]) - true0) / scale)
    B.SPLIT_CONTRACTION = None
    print(f"n={n} d={d} t={t} {mode}: err/sum|terms| f32 {out[0]:.2e} split {out[1]:.2e}   (cancellation {scale/abs(true0):.1e})", flush=True)
for n, d in ((900, 5), (2048, 3)):
    for t in (1, 4, 16, 65):
        for mode in ("abs", "randn", "scaled"):
            run(n, d, t, mode)
