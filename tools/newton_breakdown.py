"""Where one Newton step of the 512^3 Poisson configuration spends its wall time (host view)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples", "poisson"))
import odil_amd as odil
from odil_amd import gmg, ops, util
import poisson

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
args = poisson.parse_args(["--ndim", "3", "--N", str(n), "--multigrid", "0", "--linsolver", "multigrid", "--linsolver_tol", "1e-10"])
problem, state = poisson.make_problem(args)
odil.util.set_log_file(open(os.devnull, "w"))
args.epoch_start, args.epochs = 0, 0
odil.util.optimize(args, "newton", problem, state, None)
torch.cuda.synchronize()

def tick(label, t0):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("%-28s %8.2f ms" % (label, (t1 - t0) * 1e3))
    return t1

t = time.perf_counter()
ev = problem._fused
(field,) = state.fields.values()
u = field.array.contiguous()
r, _ = ops.poisson_residual(u, ev.rhs, ev.h2, fu=ev.fu, loss=ev.loss)
t = tick("residual", t)
solver = gmg.PoissonGMG(ev.cshape, ev.h2, ev.dtype, ev.device)
t = tick("solver construct", t)
solver.coarse_inverse()
t = tick("coarse inverse", t)
b = ops.scale(r, -1.0, out=r)
t = tick("negate", t)
status = {}
delta = solver.solve(b, tol=1e-10, maxiter=60, status=status, copy=False)
t = tick("solve (%d cycles)" % status["niter"], t)
domain = problem.domain
packed = domain.pack_state(state)
t = tick("pack_state", t)
domain.unpack_state(packed + delta.reshape(-1), state)
t = tick("unpack_state(packed+delta)", t)
loss = problem.eval_loss_grad_device(state)[0]
t = tick("eval_loss_grad", t)
print("loss", float(loss))
t = time.perf_counter()
delta = solver.solve(b, tol=1e-10, maxiter=60, status=status, copy=False)
t = tick("solve again (%d cycles)" % status["niter"], t)

# the same step through the public driver, on a fresh problem (first-use costs included)
import cProfile, pstats
problem2, state2 = poisson.make_problem(args)
args.epoch_start, args.epochs = 0, 0
odil.util.optimize(args, "newton", problem2, state2, None)
torch.cuda.synchronize()
args.epochs = 1
pr = cProfile.Profile()
t = time.perf_counter()
pr.enable()
odil.util.optimize(args, "newton", problem2, state2, None)
torch.cuda.synchronize()
pr.disable()
t = tick("util.optimize(newton, 1 epoch)", t)
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
