"""Epoch time of the 512^3 Poisson Adam loop as a function of how long the process has been running:
blocks of 5 epochs between HIP events from the first epoch on, and the host's enqueue time of the same blocks."""
import sys, time, torch
sys.path.insert(0, '.')
from odil_amd.poisson_path import PoissonMultigridAdam
dev = torch.device('cuda:0')
p = PoissonMultigridAdam(3, 512, dtype=torch.float64, device=dev)
nb, per = 80, 5
ev = [torch.cuda.Event(enable_timing=True) for _ in range(nb + 1)]
host = []
torch.cuda.synchronize()
ev[0].record()
for b in range(nb):
    t0 = time.perf_counter()
    for _ in range(per):
        p.epoch()
    host.append((time.perf_counter() - t0) / per * 1e3)
    ev[b + 1].record()
torch.cuda.synchronize()
gpu = [ev[b].elapsed_time(ev[b + 1]) / per for b in range(nb)]
for b in list(range(0, 12)) + list(range(12, nb, 8)):
    print("epochs %3d-%3d: gpu %.3f ms / epoch   host enqueue %.3f ms / epoch" % (b * per, b * per + per - 1, gpu[b], host[b]))
