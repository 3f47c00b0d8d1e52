"""k_poisson_adjoint_tile alone at slab-like shapes: python3 tools/mb_adjoint_tile.py"""
import sys, torch
sys.path.insert(0, '.')
from odil_amd import ops
dev = torch.device('cuda:0')
def run(nz, cut, reps=10):
    shp = (nz, 512, 512); cs = (nz // 2, 256, 256)
    fu = torch.randn(shp, dtype=torch.float64, device=dev)
    mk = lambda s: torch.zeros(s, dtype=torch.float64, device=dev)
    x0, m0, v0, g1, x1, m1, v1 = mk(shp), mk(shp), mk(shp), mk(cs), mk(cs), mk(cs), mk(cs)
    h2 = [1.0 / 512**2] * 3
    f = lambda: ops.poisson_adjoint_transpose(fu, h2, 1e-8, g1, adam0=(x0, m0, v0), adam1=(x1, m1, v1), alpha=1e-3,
                                              one_minus_b1=0.1, one_minus_b2=0.001, eps=1e-7, cut=cut)
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    print("nz %d cut %s: %.3f ms (%.4f us / plane)" % (nz, cut, a.elapsed_time(b) / reps, a.elapsed_time(b) / reps / nz * 1e3))
for nz, cut in [(512, (False, False)), (512, (True, True)), (514, (False, True)), (516, (True, True)), (520, (True, True)), (528, (True, True)), (544, (True,True))]:
    run(nz, cut)
