"""Times the P^T chain of the 512^3 f64 headline (first level + Adam) for the loaded library."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odil_amd import ops
from odil_amd.poisson_path import PoissonMultigridAdam
dev = torch.device('cuda:0')
p = PoissonMultigridAdam(3, 512, dtype=torch.float64, device=dev)
for _ in range(5): p.epoch()
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): p.epoch()
b.record(); torch.cuda.synchronize()
print("epoch %.3f ms" % (a.elapsed_time(b) / 20))
