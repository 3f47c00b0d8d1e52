cd $GRAFT_REPO_ROOT
ODIL_HIP_LIB=$PWD/odil_amd/libodil_hip_x_PHASES.so timeout 300 python3 - <<'PY'
import sys, torch
sys.path.insert(0, '.')
from odil_amd import ops
dev = torch.device('cuda:0')
shp = (512, 512, 512); cs = (256, 256, 256)
fu = torch.randn(shp, dtype=torch.float64, device=dev)
mk = lambda s: torch.zeros(s, dtype=torch.float64, device=dev)
x0, m0, v0, g1, x1, m1, v1 = mk(shp), mk(shp), mk(shp), mk(cs), mk(cs), mk(cs), mk(cs)
h2 = [1.0 / 512**2] * 3
kw = dict(alpha=1e-3, one_minus_b1=0.1, one_minus_b2=0.001, eps=1e-7)
for i in range(2):
    ops.poisson_adjoint_transpose(fu, h2, 1e-8, g1, adam0=(x0, m0, v0), adam1=(x1, m1, v1), **kw)
    torch.cuda.synchronize()
    print("--- call", i, flush=True)
PY
