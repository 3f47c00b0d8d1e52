"""Per-rank epoch time of the slab-decomposed traced tracer workload with the ranks emulated on ONE GPU
(device copies instead of xGMI messages) next to the undivided single-GPU epoch of the same per-rank size:
python3 tools/slab_traced_emulated.py [world Nt Nx]   (global grid (Nt, world*Nx, Nx, Nx), veltracer3d, f32)"""
import argparse, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples", "velocity_from_tracer"))
import odil_amd as odil
import veltracer3d
from odil_amd.slab import run_lockstep
from odil_amd.slab_traced import SlabTracedAdam, shape_state
world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 32
nx = int(sys.argv[3]) if len(sys.argv) > 3 else 128
odil.util.set_log_file(open(os.devnull, "w"))
ev = lambda: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
args = veltracer3d.parse_args(["--Nt", str(nt), "--Nx", str(world * nx), "--Ny", str(nx), "--Nz", str(nx)])
problem, state = veltracer3d.make_problem(args)
ranks = [SlabTracedAdam(problem, state, r, world, lr=0.01) for r in range(world)]
del state
run_lockstep(ranks, 2); torch.cuda.synchronize()
class T:
    def __init__(self): self.pairs = {}
    def section(self, name):
        a, b = ev(); self.pairs.setdefault(name, []).append((a, b)); return a, b
t = T()
a, b = ev(); a.record(); run_lockstep(ranks, 5, t); b.record(); torch.cuda.synchronize()
print("slab x%d emulated, %dx%dx%dx%d per rank: %.3f ms per epoch and rank" % (world, nt, nx, nx, nx, a.elapsed_time(b) / 5 / world))
print({k: round(sum(x.elapsed_time(y) for x, y in v) / 5, 3) for k, v in t.pairs.items()})
del ranks; torch.cuda.empty_cache()
args = veltracer3d.parse_args(["--Nt", str(nt), "--Nx", str(nx)])
problem, state = veltracer3d.make_problem(args)
args.epoch_start, args.epochs = 0, 2
odil.util.optimize(args, "adam", problem, state, None); torch.cuda.synchronize()
args.epochs = 40  # (the call's own set-up -- moment arrays, packed vector -- spread over enough epochs)
a, b = ev(); a.record(); odil.util.optimize(args, "adam", problem, state, None); b.record(); torch.cuda.synchronize()
print("single GPU %dx%d^3: %.3f ms per epoch" % (nt, nx, a.elapsed_time(b) / 40))
