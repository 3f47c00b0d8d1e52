"""k_poisson_adjoint_tile alone at 512^3 with / without the coarse level's Adam: ODIL_HIP_LIB=<lib> python3 tools/mb_adjoint_one.py"""
import sys, torch
sys.path.insert(0, '.')
from odil_amd import ops
dev = torch.device('cuda:0')
shp = (512, 512, 512); cs = (256, 256, 256)
fu = torch.randn(shp, dtype=torch.float64, device=dev)
mk = lambda s: torch.zeros(s, dtype=torch.float64, device=dev)
x0, m0, v0, g1, x1, m1, v1 = mk(shp), mk(shp), mk(shp), mk(cs), mk(cs), mk(cs), mk(cs)
h2 = [1.0 / 512**2] * 3
def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
kw = dict(alpha=1e-3, one_minus_b1=0.1, one_minus_b2=0.001, eps=1e-7)
full = lambda: ops.poisson_adjoint_transpose(fu, h2, 1e-8, g1, adam0=(x0, m0, v0), adam1=(x1, m1, v1), **kw)
no1 = lambda: ops.poisson_adjoint_transpose(fu, h2, 1e-8, g1, adam0=(x0, m0, v0), **kw)
res = []
for _ in range(3):
    res.append((t(full), t(no1)))
print("full / without coarse Adam (ms):", "  ".join("%.3f / %.3f" % r for r in res))
