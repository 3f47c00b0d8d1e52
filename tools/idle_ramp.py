"""Does the slow start of the 512^3 epoch come back after the GPU has idled?  30 epochs, an idle gap, 15 epochs timed
one by one -- for several gap lengths."""
import sys, time, torch
sys.path.insert(0, '.')
from odil_amd.poisson_path import PoissonMultigridAdam
dev = torch.device('cuda:0')
p = PoissonMultigridAdam(3, 512, dtype=torch.float64, device=dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(64)]
for e in ev: e.record()
torch.cuda.synchronize()
def timed(n):
    for i in range(n):
        ev[i].record(); p.epoch()
    ev[n].record(); torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print("from process start:", " ".join("%.2f" % t for t in timed(20)))
for gap in (0.0, 0.002, 0.02, 0.2, 2.0):
    for _ in range(30): p.epoch()
    torch.cuda.synchronize()
    time.sleep(gap)
    print("after 30 epochs + %.3f s idle:" % gap, " ".join("%.2f" % t for t in timed(12)))
