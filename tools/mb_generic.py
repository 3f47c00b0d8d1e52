"""Throughput of the generic (any-layout) kernels at large sizes: which ones need a fast path."""
import sys, torch
sys.path.insert(0, '.')
from odil_amd import ops
dev = torch.device('cuda:0')
def bench(f, n=5):
    for _ in range(2): f()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
N = 256
for dt in [torch.float64, torch.float32]:
    es = 8 if dt == torch.float64 else 4
    u = torch.randn((N, N, N), dtype=dt, device=dev)
    nb = u.numel() * es
    ms = bench(lambda: ops.field_gather(u, "ccc", (0, 1, -1)))
    print(dt, "field_gather ccc shift   %.3f ms %.2f TB/s" % (ms, 2 * nb / ms / 1e9))
    ms = bench(lambda: ops.field_scatter(u, (N, N, N), "ccc", (0, 1, -1)))
    print(dt, "field_scatter ccc shift  %.3f ms %.2f TB/s" % (ms, 2 * nb / ms / 1e9))
    ms = bench(lambda: ops.mean_reduce(u))
    print(dt, "mean_reduce              %.3f ms %.2f TB/s" % (ms, nb / ms / 1e9))
    shifts = [(0, 0, 0), (-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1)]
    co = torch.randn((7, N, N, N), dtype=dt, device=dev)
    ms = bench(lambda: ops.stencil_apply(co, shifts, u))
    print(dt, "stencil_apply M x        %.3f ms %.2f TB/s" % (ms, 9 * nb / ms / 1e9))
    ms = bench(lambda: ops.stencil_apply(co, shifts, u, transpose=True))
    print(dt, "stencil_apply M^T x      %.3f ms %.2f TB/s" % (ms, 9 * nb / ms / 1e9))
    un = torch.randn((N + 1, N, N), dtype=dt, device=dev)
    ms = bench(lambda: ops.field_gather(un, "ncc", (-1, 0, 1)))
    print(dt, "field_gather ncc shift   %.3f ms %.2f TB/s" % (ms, 2 * un.numel() * es / ms / 1e9))
    c = torch.randn((N // 2 + 1, N // 2, N // 2), dtype=dt, device=dev)
    ms = bench(lambda: ops.interp_add(c, "ncc"))
    print(dt, "interp_add ncc           %.3f ms %.2f TB/s" % (ms, (N + 1) * N * N * es / ms / 1e9))
    gf = torch.randn((N + 1, N, N), dtype=dt, device=dev)
    ms = bench(lambda: ops.interp_adj(gf, "ncc", tuple(c.shape)))
    print(dt, "interp_adj ncc           %.3f ms %.2f TB/s" % (ms, (N + 1) * N * N * es / ms / 1e9))
    cn = torch.randn((N // 2 + 1,) * 3, dtype=dt, device=dev)
    ms = bench(lambda: ops.interp_add(cn, "nnn"))
    print(dt, "interp_add nnn (generic) %.3f ms %.2f TB/s" % (ms, (N + 1) ** 3 * es / ms / 1e9))
    gn = torch.randn((N + 1,) * 3, dtype=dt, device=dev)
    ms = bench(lambda: ops.interp_adj(gn, "nnn", tuple(cn.shape)))
    print(dt, "interp_adj nnn (generic) %.3f ms %.2f TB/s" % (ms, (N + 1) ** 3 * es / ms / 1e9))
