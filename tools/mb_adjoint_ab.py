"""k_poisson_adjoint_tile alone (finest-level Adam + first P^T fused), several shapes:
ODIL_HIP_LIB=<lib> python3 tools/mb_adjoint_ab.py      (border tiles: 23 % of the tiles at 512^2 planes, 12 % at 1024^2)"""
import sys, torch
sys.path.insert(0, '.')
from odil_amd import ops
dev = torch.device('cuda:0')
def run(shp, reps=10):
    cs = tuple(s // 2 for s in shp)
    fu = torch.randn(shp, dtype=torch.float64, device=dev)
    mk = lambda s: torch.zeros(s, dtype=torch.float64, device=dev)
    x0, m0, v0, g1, x1, m1, v1 = mk(shp), mk(shp), mk(shp), mk(cs), mk(cs), mk(cs), mk(cs)
    h2 = [1.0 / 512**2] * 3
    f = lambda: ops.poisson_adjoint_transpose(fu, h2, 1e-8, g1, adam0=(x0, m0, v0), adam1=(x1, m1, v1), alpha=1e-3,
                                              one_minus_b1=0.1, one_minus_b2=0.001, eps=1e-7)
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    n = shp[0] * shp[1] * shp[2]
    print("%s: %.3f ms  %.3f ps/cell  %.2f TB/s on 8.125 words" % (shp, ms, ms * 1e9 / n, 8.125 * 8 * n / ms / 1e9))
for shp in [(512, 512, 512), (128, 1024, 1024), (32, 2048, 2048), (512, 512, 512)]:
    run(shp)
