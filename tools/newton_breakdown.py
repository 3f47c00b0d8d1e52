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
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
# host time from the call to the FIRST launch of the step (the GPU idles meanwhile) and between solve phases
import odil_amd.util as U
real = U._poisson_newton_step
marks = []
def timed(problem, state, args, status):
    marks.append(("enter step", time.perf_counter()))
    out = real(problem, state, args, status)
    marks.append(("leave step (host)", time.perf_counter()))
    return out
U._poisson_newton_step = timed
state2.fields[list(state2.fields)[0]].array.zero_()
torch.cuda.synchronize()
t0 = time.perf_counter()
odil.util.optimize(args, "newton", problem2, state2, None)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
for name, t in marks:
    print("%-24s +%.3f ms" % (name, 1e3 * (t - t0)))
print("optimize returned (host)  +%.3f ms; GPU done +%.3f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t0)))
