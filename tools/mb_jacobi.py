import torch, time, sys
sys.path.insert(0, '.')
from odil_amd import ops
dev = torch.device('cuda:0')
N = 512
n = N**3
h2 = [1.0 / N**2] * 3
def bench(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for pad in [0, 256, 4096, 65536 + 256, 1 << 20, (1 << 20) + 4096 + 256, 3 * (1 << 19) + 8192]:
    pool = torch.zeros(3 * n + 3 * pad // 8 + 16, dtype=torch.float64, device=dev)
    views = []
    off = 0
    for k in range(3):
        views.append(pool[off:off + n].view(N, N, N))
        off += n + pad // 8
    u, r, o = views
    u.normal_(); r.normal_()
    tj = bench(lambda: ops.poisson_jacobi(u, r, h2, 0.8, o))
    loss = torch.zeros((), dtype=torch.float64, device=dev)
    tr = bench(lambda: ops.poisson_residual(u, r, h2, fu=o, loss=loss))
    print("pad %9d B: jacobi %.3f ms (%.2f TB/s)  residual %.3f ms" % (pad, tj, 3 * n * 8 / tj / 1e9, tr))
    del pool, views, u, r, o
