"""The same Newton step of config 4b timed the way bench.py's other_configs times it (HIP events around two back-to-back
steps) and the way tools/newton_ab.py does (host clock around one synchronised step), in one process."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples", "poisson"))
import odil_amd as odil, poisson
odil.util.set_log_file(open(os.devnull, "w"))
a = poisson.parse_args(["--ndim", "3", "--N", "512", "--multigrid", "0", "--linsolver", "multigrid", "--linsolver_tol", "1e-10"])
problem, state = poisson.make_problem(a)
a.epoch_start, a.epochs = 0, 1
def step():
    for f in state.fields.values():
        f.array.zero_()
    odil.util.optimize(a, "newton", problem, state, None)
for _ in range(2): step()
torch.cuda.synchronize()
for rnd in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); step(); step(); e1.record(); torch.cuda.synchronize()
    ev = e0.elapsed_time(e1) / 2
    ts = []
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    t0 = time.perf_counter(); step(); step(); torch.cuda.synchronize(); bb = 1e3 * (time.perf_counter() - t0) / 2
    print("round %d: events over two back-to-back steps %.2f | host clock, one step at a time %.2f %.2f | host clock over two back-to-back %.2f" % (rnd, ev, ts[0], ts[1], bb), flush=True)
