"""The space part of the 4-D tracer's transposes alone: P^T of a (129, 36, 256, 256) float array on the layout '.ccc'
(k_interp_adj_tile<float, 2>, one volume per blockIdx.y), and the double 512^3 -> 256^3 of the Poisson chain.
ODIL_HIP_LIB=<lib> python3 tools/mb_adj_tile_f32.py"""
import sys, torch
sys.path.insert(0, '.')
from odil_amd import ops
dev = torch.device('cuda:0')
def t(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
g = torch.randn((129, 36, 256, 256), dtype=torch.float32, device=dev)
out = torch.empty((129, 18, 128, 128), dtype=torch.float32, device=dev)
ms = t(lambda: ops.interp_adj(g, ".ccc", (129, 18, 128, 128), out=out))
print("float  (129, 36, 256, 256) '.ccc': %.3f ms  %.2f TB/s" % (ms, (g.numel() + out.numel()) * 4 / ms / 1e9))
g = torch.randn((512, 512, 512), dtype=torch.float64, device=dev)
out = torch.empty((256, 256, 256), dtype=torch.float64, device=dev)
ms = t(lambda: ops.interp_adj(g, "ccc", (256, 256, 256), out=out))
print("double (512, 512, 512) 'ccc':      %.3f ms  %.2f TB/s" % (ms, (g.numel() + out.numel()) * 8 / ms / 1e9))
