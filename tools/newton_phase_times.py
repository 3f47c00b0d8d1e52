import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
R = sys.path[0]
sys.path.insert(0, os.path.join(R, "examples", "poisson")); sys.path.insert(0, os.path.join(R, "examples", "diffusion"))
import odil_amd as odil
from odil_amd import linsolver, gmg
odil.util.set_log_file(open(os.devnull, "w"))
which = sys.argv[1] if len(sys.argv) > 1 else "diffusion"
if which == "varcoef":
    os.environ["ODIL_NEWTON_SHORTCUT"] = "0"; os.environ["ODIL_GMG"] = "stencil"
    import poisson as ex
    args = ex.parse_args(["--ndim", "3", "--N", "512", "--multigrid", "0", "--linsolver", "multigrid", "--linsolver_tol", "1e-10"])
else:
    import diffusion as ex
    args = ex.parse_args(["--ndim", "3", "--N", "256", "--kind", "jump", "--linsolver", "multigrid", "--linsolver_tol", "1e-10"])
problem, state = ex.make_problem(args)
def sync():
    torch.cuda.synchronize(); return time.perf_counter()
for step in range(4):
    for f in state.fields.values(): f.array.zero_()
    t0 = sync()
    vector, matrix = problem.linearize_device(state)
    t1 = sync()
    st = {}
    delta = linsolver.solve(matrix, odil.ops.scale(vector.contiguous(), -1.0) if hasattr(odil, "ops") else -vector, args, st, "multigrid")
    t2 = sync()
    print(which, "step", step, "linearize %.1f ms  solve %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), st.get("niter"), st.get("method"), flush=True)
cache = problem.domain.__dict__.get("_gmg_cache", {})
for key, solver in cache.items():
    ug = solver.__dict__.get("_unit_graphs")
    print("solver", key[0], "units", None if ug is None else ug["count"], "graphs", None if ug is None else len(ug["graphs"]), "off", None if ug is None else ug["off"])
# one more solve with per-cycle timing
import odil_amd.gmg as G
solver = list(cache.values())[0]
orig = solver._cycle_unit
times = []
def timed_unit(x, b):
    t0 = sync(); y = orig(x, b); t1 = sync(); times.append((t1 - t0) * 1e3); return y
solver._cycle_unit = timed_unit
for f in state.fields.values(): f.array.zero_()
vector, matrix = problem.linearize_device(state)
t0 = sync(); st = {}
delta = linsolver.solve(matrix, -vector, args, st, "multigrid"); t1 = sync()
print("solve %.1f ms; units: %s" % ((t1 - t0) * 1e3, " ".join("%.2f" % t for t in times)))
