import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpytorch_amd as g
from gpytorch_amd import backend as B
from oracle import kernels as OK
dev = torch.device("cuda:0")
n, d = 900, 5
gen = torch.Generator().manual_seed(0)
x = torch.rand(n, d, generator=gen)
rhs0 = torch.randn(n, 4, generator=gen)
grad = torch.randn(n, 4, generator=gen)
ls0 = torch.tensor([[0.6]])
ls64 = ls0.double().requires_grad_(True)
rhs64 = rhs0.double().requires_grad_(True)
K = OK.rbf(x.double(), x.double(), ls64, x1_eq_x2=True) + torch.eye(n, dtype=torch.float64)
actual = torch.linalg.solve(K, rhs64)
actual.backward(gradient=grad.double())
for split in (False, True):
    for tol in (1e-4, 1e-6):
        B.SPLIT_CONTRACTION = split
        kern = g.kernels.RBFKernel().to(dev)
        kern.lengthscale = ls0
        rhs = rhs0.to(dev).requires_grad_(True)
        with g.settings.max_cholesky_size(0), g.settings.cg_tolerance(tol), g.settings.max_preconditioner_size(0):
            res = kern(x.to(dev), x.to(dev)).add_jitter(1.0).solve(rhs)
        res.backward(gradient=grad.to(dev))
        chain = torch.sigmoid(kern.raw_lengthscale.detach().double().cpu())
        got = float(kern.raw_lengthscale.grad.double().cpu().sum()); want = float((ls64.grad * chain).sum())
        print("split" if split else "f32", tol, "grad", got, "want", want, "rel", abs(got - want) / abs(want), "solve err", float((res.detach().double().cpu() - actual.detach()).abs().max() / actual.abs().max()), flush=True)
B.SPLIT_CONTRACTION = None
