"""A/B of solver switches on ONE box (boxes of the pool differ by several percent): the Newton step of config 4b / 4c /
the variable-coefficient route with class attributes of the multigrid solvers toggled, alternating, best of each.

    python tools/newton_ab.py [poisson|varcoef|diffusion] [N]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples", "poisson"))
sys.path.insert(0, os.path.join(ROOT, "examples", "diffusion"))
import odil_amd as odil  # noqa: E402
from odil_amd import gmg  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "poisson"
odil.util.set_log_file(open(os.devnull, "w"))
if which == "diffusion":
    import diffusion as ex
    n = sys.argv[2] if len(sys.argv) > 2 else "256"
    args = ex.parse_args(["--ndim", "3", "--N", n, "--kind", "jump", "--linsolver", "multigrid", "--linsolver_tol", "1e-10"])
else:
    if which == "varcoef":
        os.environ["ODIL_NEWTON_SHORTCUT"] = "0"
        os.environ["ODIL_GMG"] = "stencil"
    import poisson as ex
    n = sys.argv[2] if len(sys.argv) > 2 else "512"
    args = ex.parse_args(["--ndim", "3", "--N", n, "--multigrid", "0", "--linsolver", "multigrid", "--linsolver_tol", "1e-10"])
problem, state = ex.make_problem(args)
args.epoch_start, args.epochs = 0, 1


def step():
    for f in state.fields.values():
        f.array.zero_()
    seen = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    odil.util.optimize(args, "newton", problem, state, lambda s, e, p: seen.append(p.get("linsolver")))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    st = next((s for s in seen if s and "niter" in s), {})
    return dt, st.get("niter")


VARIANTS = {
    "pairs off, tail off": dict(pair_min_cells=10**18, tail_max_cells=0),
    "pairs off, tail on ": dict(pair_min_cells=10**18, tail_max_cells=8192),
    "pairs >= 128^3, tail": dict(pair_min_cells=128**3, tail_max_cells=8192),
    "pairs >= 64^3, tail": dict(pair_min_cells=64**3, tail_max_cells=8192),
    "pairs >= 32^3, tail": dict(pair_min_cells=32**3, tail_max_cells=8192),
}
if len(sys.argv) > 3 and sys.argv[3] == "zero":
    VARIANTS = {"zero iterates written and read": dict(zero_start=False), "zero start not read   ": dict(zero_start=True)}
best = {k: (1e9, None) for k in VARIANTS}
for rnd in range(4):
    for name, attrs in VARIANTS.items():
        for k, v in attrs.items():
            setattr(gmg.PoissonGMG, k, v)
        problem.domain.__dict__.pop("_poisson_gmg", None)  # (solvers are kept with the domain: rebuild under the new switches)
        getattr(problem, "_fused", None) is not None and problem._fused.__dict__.pop("_gmg", None)
        step()
        dt, it = min(step() for _ in range(2))
        if dt < best[name][0]:
            best[name] = (dt, it)
for name, (dt, it) in best.items():
    print("{} {}: {:.2f} ms per Newton step, {} cycles".format(which, name, dt, it), flush=True)
