"""Host enqueue time of the single-GPU Poisson epoch against its GPU time (is the host ever the limit?)."""
import sys, time, torch
sys.path.insert(0, '.')
from odil_amd.poisson_path import PoissonMultigridAdam
import bench
dev = torch.device('cuda:0')
for N in (512, 256):
    p = PoissonMultigridAdam(3, N, dtype=torch.float64, device=dev)
    for _ in range(10): p.epoch()
    torch.cuda.synchronize()
    for timers in (None, bench.Timers()):
        t0 = time.perf_counter()
        for _ in range(20): p.epoch(timers)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("N %d timers %s: enqueue %.3f ms / epoch, wall %.3f ms / epoch" % (N, timers is not None, (t1 - t0) * 50, (t2 - t0) * 50))
