"""Does replaying the 512^3 epoch as one hipGraph beat the eager launches? (25 dependent kernels)"""
import sys, torch, numpy as np
sys.path.insert(0, '.')
from odil_amd.poisson_path import PoissonMultigridAdam
dev = torch.device('cuda:0')
p = PoissonMultigridAdam(3, 512, dtype=torch.float64, device=dev)
for _ in range(5): p.epoch()
torch.cuda.synchronize()
def timeit(f, n=20):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print("eager  %.3f ms" % timeit(p.epoch))
alpha = torch.zeros(1, dtype=torch.float64, device=dev)
omb1, omb2 = 1 - p.b1, 1 - p.b2
def body():
    p.ev.loss_grad_arrays(p.w, None, adam=(p.mw, p.vw, alpha, omb1, omb2, p.eps))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    body()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
host = torch.zeros(1, dtype=torch.float64).pin_memory()
def replay():
    p.t += 1
    t = np.float64(p.t)
    host[0] = p.lr * np.sqrt(1 - p.b2**t) / (1 - p.b1**t)
    alpha.copy_(host, non_blocking=True)
    g.replay()
for _ in range(3): replay()
print("graph  %.3f ms" % timeit(replay))
print("loss", p.last_loss())
