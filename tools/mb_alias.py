"""Do streams whose base addresses differ by exactly 2^30 bytes (512^3 doubles) collide in the memory system?  The pair of
Jacobi sweeps and the single sweep with x, b, out carved from ONE buffer at offsets k (2^30 + pad) for several pads."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odil_amd import ops
dev = torch.device("cuda:0")
n = 512
size = n ** 3
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
h2 = [1.0 / n**2] * 3
for pad_bytes in (0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, 3 << 20, 37 * 4096):
    pad = pad_bytes // 8
    big = torch.zeros(3 * (size + pad) + 16, dtype=torch.float64, device=dev)
    x = big[0:size].view(n, n, n); b = big[size + pad:2 * size + pad].view(n, n, n); out = big[2 * (size + pad):2 * (size + pad) + size].view(n, n, n)
    x.normal_(); b.normal_()
    t1 = timeit(lambda: ops.poisson_jacobi(x, b, h2, 0.9, out))
    t2 = timeit(lambda: ops.poisson_jacobi2(x, b, h2, 0.9, 0.6, out))
    r, _ = ops.poisson_residual(x, b, h2, fu=out)
    t3 = timeit(lambda: ops.poisson_residual(x, b, h2, fu=out))
    print("pad %8d B: jacobi %.3f ms  jacobi2 %.3f ms  residual %.3f ms" % (pad_bytes, t1, t2, t3), flush=True)
    del big, x, b, out
    torch.cuda.empty_cache()
