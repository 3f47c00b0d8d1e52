"""Tracer velocity 128x256x256 (BASELINE config 5): eager epochs against hipGraph replay, steady state."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples", "velocity_from_tracer"))
import odil_amd as odil
import veltracer
odil.util.set_log_file(open(os.devnull, "w"))
args = veltracer.parse_args(["--Nt", "128", "--Nx", "256", "--Ny", "256"])
problem, state = veltracer.make_problem(args)
marks = []
def cb(s, e, p):
    if e in (100, 300):
        torch.cuda.synchronize(); marks.append(time.perf_counter())
args.epoch_start, args.epochs = 0, 300
odil.util.optimize(args, "adam", problem, state, cb)
print("ODIL_GRAPH=%s: %.3f ms / epoch over epochs 100..300" % (os.environ.get("ODIL_GRAPH", "auto"), (marks[1] - marks[0]) / 200 * 1e3))
