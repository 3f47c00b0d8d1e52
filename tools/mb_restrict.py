import torch, sys
sys.path.insert(0, '.')
from odil_amd import ops
dev = torch.device('cuda:0')
for shape, loc in [((512, 512, 512), "ccc"), ((4096, 4096), "cc")]:
    for dt in [torch.float64, torch.float32]:
        u = torch.randn(shape, dtype=dt, device=dev)
        for _ in range(3): ops.restrict_to_coarser(u, loc)
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): ops.restrict_to_coarser(u, loc)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        nbytes = u.numel() * u.element_size() * (1 + 0.5 ** len(shape))
        print(loc, shape, dt, "%.3f ms  %.2f TB/s" % (ms, nbytes / ms / 1e9))
for shape, loc in [((512, 512, 512), "ccc"), ((4096, 4096), "cc")]:
    for dt in [torch.float64, torch.float32]:
        cs = tuple(n // 2 for n in shape)
        g = torch.randn(cs, dtype=dt, device=dev)
        for _ in range(3): ops.restrict_adj(g, loc, shape)
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): ops.restrict_adj(g, loc, shape)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        nbytes = g.numel() * g.element_size() * (1 + 2 ** len(shape))
        print("restrict_adj", loc, shape, dt, "%.3f ms  %.2f TB/s" % (ms, nbytes / ms / 1e9))
