"""What ends the slow start after idle: 30 epochs, 0.2 s idle, then 80 ms of (a) fills, (b) read-modify-write passes,
(c) nothing, then 10 epochs timed one by one."""
import sys, time, torch
sys.path.insert(0, '.')
from odil_amd.poisson_path import PoissonMultigridAdam
dev = torch.device('cuda:0')
p = PoissonMultigridAdam(3, 512, dtype=torch.float64, device=dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(32)]
for e in ev: e.record()
torch.cuda.synchronize()
buf = torch.zeros(256 << 20, dtype=torch.float32, device=dev)
def timed(n):
    for i in range(n):
        ev[i].record(); p.epoch()
    ev[n].record(); torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
def spin(kind, ms=80):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(8):
            if kind == "fill": buf.fill_(1.0)
            else: buf.add_(1.0)
        torch.cuda.synchronize()
for kind in ("none", "fill", "rmw", "none", "rmw"):
    for _ in range(30): p.epoch()
    torch.cuda.synchronize(); time.sleep(0.2)
    if kind != "none": spin(kind)
    print("%-5s" % kind, " ".join("%.2f" % t for t in timed(10)))
