"""Per-rank epoch time of the slab decomposition with the ranks emulated on ONE GPU (device copies instead of
xGMI messages): what the kernels of a rank cost next to the single-GPU epoch."""
import sys, torch
sys.path.insert(0, '.')
from odil_amd.slab import SlabPoissonAdam, run_lockstep
from odil_amd.poisson_path import PoissonMultigridAdam
dev = torch.device('cuda:0')
N, world = 512, int(sys.argv[1]) if len(sys.argv) > 1 else 2
ranks = [SlabPoissonAdam(N, r, world, dtype=torch.float64, device=dev) for r in range(world)]
run_lockstep(ranks, 3)
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
class T:
    def __init__(self): self.pairs = {}
    def section(self, name):
        x, y = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.pairs.setdefault(name, []).append((x, y)); return x, y
t = T()
a.record(); run_lockstep(ranks, 10, t); b.record(); torch.cuda.synchronize()
print("slab x%d emulated: %.3f ms per epoch and rank" % (world, a.elapsed_time(b) / 10 / world))
# (sections of rank 0; "halo" brackets the other ranks' kernels of the same step as well)
print({k: round(sum(x.elapsed_time(y) for x, y in v) / 10, 3) for k, v in t.pairs.items()})
del ranks; torch.cuda.empty_cache()
p = PoissonMultigridAdam(3, N, dtype=torch.float64, device=dev)
for _ in range(3): p.epoch()
a.record()
for _ in range(10): p.epoch()
b.record(); torch.cuda.synchronize()
print("single GPU: %.3f ms per epoch" % (a.elapsed_time(b) / 10))
