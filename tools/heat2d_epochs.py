"""Runs a few Adam epochs of the traced heat2d operator (config 3 at BASELINE's shape, (t,x,y) = 256 x 512^2 f32)
for profiling: python3 tools/heat2d_epochs.py [Nt Nx epochs]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples", "heat"))
import odil_amd as odil
import heat2d
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 512
ep = int(sys.argv[3]) if len(sys.argv) > 3 else 6
args = heat2d.parse_args(["--Nt", str(nt), "--Nx", str(nx), "--Ny", str(nx), "--infer_k", "1", "--imposed", "stripe"])
odil.util.set_log_file(open(os.devnull, "w"))
problem, state = heat2d.make_problem(args)
args.epoch_start, args.epochs = 0, 2
odil.util.optimize(args, "adam", problem, state, None)
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
args.epochs = ep
a.record(); odil.util.optimize(args, "adam", problem, state, None); b.record(); torch.cuda.synchronize()
print("heat2d %dx%dx%d: %.3f ms / epoch" % (nt, nx, nx, a.elapsed_time(b) / ep))
