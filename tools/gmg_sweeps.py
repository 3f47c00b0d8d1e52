"""V(nu1, nu2) choices for the 512^3 Newton solve: cycles x cost."""
import sys, time, torch
sys.path.insert(0, '.')
from odil_amd import gmg, ops
dev = torch.device('cuda:0')
N = 512
h2 = [1.0 / N**2] * 3
torch.manual_seed(0)
b = torch.randn((N, N, N), dtype=torch.float64, device=dev)
for nu1, nu2 in [(2, 2), (3, 3), (2, 1), (1, 2), (3, 2), (4, 4)]:
    s = gmg.PoissonGMG((N, N, N), h2, torch.float64, dev, nu1=nu1, nu2=nu2)
    st = {}
    s.solve(b, tol=1e-10, maxiter=60, status=st, copy=False)
    torch.cuda.synchronize(); t = time.perf_counter()
    s.solve(b, tol=1e-10, maxiter=60, status=st, copy=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("V(%d,%d): %2d cycles, %.1f ms, residual %.2e" % (nu1, nu2, st["niter"], dt * 1e3, st["residual"]))
    del s
    torch.cuda.empty_cache()
