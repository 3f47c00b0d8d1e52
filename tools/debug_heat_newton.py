"""Where the time of a Newton step of heat with the conductivity network goes: linearize, solve (method, status)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples", "heat"))
import odil_amd as odil
import heat as ex
import argparse
odil.util.set_log_file(open(os.devnull, "w"))
for nt, nx in ((64, 128), (256, 512)):
    base = ["--multigrid", "0", "--double", "1", "--infer_k", "1", "--imposed", "stripe", "--kxreg", "0.1", "--ktreg", "0.05"]
    args = ex.parse_args(["--Nt", str(nt), "--Nx", str(nx)] + base)
    problem, state = ex.make_problem(args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    vector, op = problem.linearize_device(state)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(nt, nx, "linearize %.3f s" % (t1 - t0), "blocks", [(k, kind) for _, _, kind, k, _ in op.blocks][:12], flush=True)
    for dd in (0.1, 0.0):
        status = dict()
        largs = argparse.Namespace(linsolver_maxiter=200, linsolver_damp=0, linsolver_dampdiag=dd, linsolver_tol=1e-10)
        t0 = time.perf_counter()
        x = odil.linsolver.solve(op, -vector, largs, status, "direct")
        torch.cuda.synchronize()
        print("  dampdiag", dd, "solve %.3f s" % (time.perf_counter() - t0), {k: v for k, v in status.items()}, flush=True)
