import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
R = sys.path[0]
sys.path.insert(0, os.path.join(R, "examples", "poisson")); sys.path.insert(0, os.path.join(R, "examples", "diffusion"))
import odil_amd as odil
from odil_amd import linsolver, gmg
odil.util.set_log_file(open(os.devnull, "w"))
which = sys.argv[1] if len(sys.argv) > 1 else "diffusion"
if which not in ("poisson", "varcoef", "diffusion"):
    sys.exit("usage: newton_phase_times.py poisson|varcoef|diffusion")
if which == "poisson":
    import poisson as ex
    args = ex.parse_args(["--ndim", "3", "--N", "512", "--multigrid", "0", "--linsolver", "multigrid", "--linsolver_tol", "1e-10"])
elif which == "varcoef":
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
