"""One Newton step of the heat example (implicit in time, network unknowns) at several sizes: which solver runs, time.
python3 tools/heat_newton.py"""
import sys, time, torch, os
sys.path.insert(0, '.'); sys.path.insert(0, 'examples/heat')
import odil_amd as odil
import heat
for nt, nx in ((32, 64), (64, 128), (128, 256), (256, 512)):
    args = heat.parse_args(["--Nt", str(nt), "--Nx", str(nx), "--optimizer", "newton", "--multigrid", "0", "--double", "1",
                            "--epochs", "1", "--linsolver", "direct", "--infer_k", "0"] if "--infer_k" in open("examples/heat/heat.py").read()
                           else ["--Nt", str(nt), "--Nx", str(nx), "--optimizer", "newton", "--multigrid", "0", "--double", "1", "--epochs", "1", "--linsolver", "direct"])
    problem, state = heat.make_problem(args)
    odil.util.set_log_file(open(os.devnull, "w"))
    args.epoch_start, args.epochs = 0, 1
    try:
        odil.util.optimize_newton(args, problem, state)  # first step: allocations, library start-up
    except Exception as e:
        pass
    torch.cuda.synchronize(); t0 = time.perf_counter()
    try:
        odil.util.optimize_newton(args, problem, state)
        ok = "ok"
    except Exception as e:
        ok = "FAILED: %s" % str(e)[:150]
    torch.cuda.synchronize()
    loss = float(problem.eval_loss_grad(state)[0])
    print("heat %dx%d newton step: %.3f s  loss after %.3e  %s" % (nt, nx, time.perf_counter() - t0, loss, ok), flush=True)
