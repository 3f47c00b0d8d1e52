"""Row-marching float P^T (csrc/mg_march.hip: k_interp_adj_rows) against the kernels it replaces (ODIL_ADJ_ROWS=0) and
against float64; timings at the shapes of config 5 (one rank) and of the tracer with three space dimensions."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odil_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def run(g, loc, cshape, rows, **kw):
    os.environ["ODIL_ADJ_ROWS"] = "1" if rows else "0"
    return ops.interp_adj(g, loc, cshape, **kw)


def check(loc, cshape):
    torch.manual_seed(sum(cshape))
    fshape = ops.fine_shape(cshape, loc)
    g64 = torch.randn(fshape, dtype=torch.float64, device=dev)
    g = g64.float()
    ref64 = ops.interp_adj(g64, loc, cshape)
    old, new = run(g, loc, cshape, False), run(g, loc, cshape, True)
    scale = float(ref64.abs().max())
    e_old, e_new = float((old.double() - ref64).abs().max()) / scale, float((new.double() - ref64).abs().max()) / scale
    same = bool((old == new).all())
    print(loc, cshape, "old %.2e new %.2e" % (e_old, e_new), "identical" if same else "", flush=True)
    assert e_new < 2e-6, (loc, cshape, e_new)
    o2, s2 = run(g, loc, cshape, True, scale=0.5)
    assert torch.equal(o2, new) and torch.equal(s2, 0.5 * new)


def timing(loc, cshape, reps=5):
    fshape = ops.fine_shape(cshape, loc)
    g = torch.randn(fshape, dtype=torch.float32, device=dev)
    out = torch.empty(cshape, dtype=torch.float32, device=dev)
    for rows in (False, True, False, True):
        run(g, loc, cshape, rows, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            run(g, loc, cshape, rows, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(loc, cshape, "rows" if rows else "tile", "%.3f ms  %.2f TB/s on the fine array" % (dt * 1e3, g.numel() * 4 / dt / 1e12),
              flush=True)


for loc, cshape in [("ccc", (4, 4, 32)), ("ccc", (5, 7, 32)), ("ccc", (18, 16, 64)), ("ccc", (9, 33, 128)), ("ccc", (16, 40, 128)),
                    (".ccc", (3, 6, 12, 32)), (".ccc", (5, 18, 20, 128)), (".ccc", (2, 4, 5, 64)), ("ccc", (128, 128, 128))]:
    check(loc, cshape)
if len(sys.argv) > 1:
    timing(".ccc", (129, 18, 128, 128))
    timing(".ccc", (33, 128, 128, 128))
    timing(".ccc", (65, 9, 64, 64))
print("ok")
