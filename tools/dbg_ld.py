import sys, numpy as np, torch
sys.path.insert(0, '.')
from odil_amd import ops
dev = torch.device('cuda:0')
rng = np.random.default_rng(67)
for loc, cshape, ldx in [(".ccc", (3, 4, 4, 32), 1), (".ccc", (3, 4, 4, 32), 2), (".ccc", (3, 4, 4, 32), 0), (".ccc", (3, 8, 8, 64), 1), ("nccc", (3, 4, 4, 32), 1)]:
    g = torch.tensor(rng.standard_normal(ops.fine_shape(cshape, loc)).astype(np.float32), device=dev)
    want = ops.interp_adj(g, loc, cshape)
    vol = cshape[1] * cshape[2] * cshape[3]
    ld = vol + ldx
    base = torch.full((3 * ld + 8,), 7.0, dtype=torch.float32, device=dev)
    view = base.as_strided(cshape, (ld, cshape[2] * cshape[3], cshape[3], 1))
    ops.interp_adj(g, loc, cshape, out=view)
    d = (view - want).abs()
    print(loc, cshape, "ld", ld, "max diff per lead", [float(d[i].max()) for i in range(cshape[0])], "outside untouched", bool((base[3 * ld:] == 7).all()))
